"""Only what pointcept/models/modules.py needs from the hook package (reference: engines/hooks/default.py)."""


class HookBase:
    """Base class for hooks; PointModel subclasses it (models/modules.py:114-120 in the reference)."""

    trainer = None

    def before_train(self):
        pass

    def before_epoch(self):
        pass

    def before_step(self):
        pass

    def after_step(self):
        pass

    def after_epoch(self):
        pass

    def after_train(self):
        pass

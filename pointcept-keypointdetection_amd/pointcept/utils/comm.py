"""Process-group helpers the hooks use (reference: pointcept/utils/comm.py:23-90): thin views of torch.distributed."""
import torch.distributed as dist


def _on():
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if _on() else 1


def get_rank() -> int:
    return dist.get_rank() if _on() else 0


def is_main_process() -> bool:
    return get_rank() == 0


def synchronize():
    if _on() and dist.get_world_size() > 1:
        dist.barrier()

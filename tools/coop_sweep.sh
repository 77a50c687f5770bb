#!/bin/bash
# sweep the row limits of the workgroup-cooperative fused block kernels (env PTV3_COOP_ROWS_<C>)
run() { echo "== $*"; env "$@" python bench.py --cpu-sample 0 --no-kernel-events 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'])
"; }
run PTV3_COOP_ROWS_128=0 PTV3_COOP_ROWS_256=0 PTV3_COOP_ROWS_512=0
run PTV3_COOP_ROWS_128=16384 PTV3_COOP_ROWS_256=0 PTV3_COOP_ROWS_512=0
run PTV3_COOP_ROWS_128=0 PTV3_COOP_ROWS_256=4096 PTV3_COOP_ROWS_512=0
run PTV3_COOP_ROWS_128=16384 PTV3_COOP_ROWS_256=4096 PTV3_COOP_ROWS_512=0
run PTV3_COOP_ROWS_128=16384 PTV3_COOP_ROWS_256=4096 PTV3_COOP_ROWS_512=4096

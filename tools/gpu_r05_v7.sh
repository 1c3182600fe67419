#!/bin/bash
# round 5, visit 7: the software-pipelined block kernel (mst_tcn_set_tuning bit 7) against the default (53), same box, alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_v7; mkdir -p $O
timeout 240 python tools/bench_tcn_forms.py --forms ${FORMS:-53,181} --rounds ${ROUNDS:-3} --steps 10 --out $O/forms.json > $O/forms.log 2>&1
tail -12 $O/forms.log

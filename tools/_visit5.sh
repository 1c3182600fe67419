cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04_v5; mkdir -p $O
one() { python bench.py --precision bf16x3 --workload configs1 --steps 4 --warmup 2 --no-cpu-baseline --tcn-tuning $1 2>$O/err_$1.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); x=json.load(open('gpurun_out/bench_details.json'))['details']['headline']['roofline']
print(round(d['value'],1), round(d['ms_per_step'],3), [round(v,3) for v in x['per_block_ms']])"; }
for r in 1 2; do for f in 5 13; do echo "tuning $f: $(one $f)" >> $O/x3_duo_ab.txt; done; done
cat $O/x3_duo_ab.txt

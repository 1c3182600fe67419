cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04_v2
O=gpurun_out/r04_v2
one() { python bench.py --precision $1 --workload configs1 --steps $2 --warmup 2 --no-cpu-baseline 2>$O/ab_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); x=json.load(open('gpurun_out/bench_details.json'))['details']['headline']['roofline']
print(round(d['value'],1), round(d['ms_per_step'],3), d['roofline'].get('calib_ms'), d['roofline'].get('sclk_mhz'), [round(v,3) for v in x['per_block_ms'][10:14]])"; }
for r in 1 2; do for v in old new; do cp tools/_ab/$v.so music_mixing_style_transfer_amd/csrc/libmst_hip.so
  echo "$v bf16: $(one bf16 10)" >> $O/ab.txt; echo "$v bf16x3: $(one bf16x3 3)" >> $O/ab.txt; done; done
cat $O/ab.txt
cp tools/_ab/new.so music_mixing_style_transfer_amd/csrc/libmst_hip.so
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25 > $O/pytest_gpu.log
tail -12 $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "rc=$?"
wc -c $O/bench_driver_cmd.json; cat $O/bench_driver_cmd.json
cp gpurun_out/bench_details.json $O/bench_details.json
timeout 600 python tools/prof_cli.py > $O/prof_cli.txt 2>&1; head -12 $O/prof_cli.txt

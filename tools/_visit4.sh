cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$(pwd); O=$R/gpurun_out/r04_v4; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "normalizer or normalize or cli or prefetch or fx" 2>&1 | tail -5 > $O/pytest_nz.log; cat $O/pytest_nz.log
timeout 600 python tools/prof_normalizer_kernels.py 3 > $O/nz_times.txt 2>&1; tail -3 $O/nz_times.txt
timeout 600 python tools/bench_cli.py --precision bf16 > $O/bench_cli_1song.json 2>$O/bench_cli.err; tail -c 400 $O/bench_cli_1song.json
python - > $O/f2f.json 2>>$O/bench_cli.err <<'PY'
import sys, json, contextlib
sys.path.insert(0, "tools")
import bench_cli
with contextlib.redirect_stdout(sys.stderr):
    r = bench_cli.run(180.0, "bf16", songs=2)
print(json.dumps(r))
PY
cat $O/f2f.json
timeout 600 python tools/prof_cli.py > $O/prof_cli.txt 2>&1; grep -n "dataset item\|normalize_audio\|effect\|warm pass" $O/prof_cli.txt

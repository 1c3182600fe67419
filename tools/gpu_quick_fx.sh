#!/bin/bash
# FX / normaliser / file-to-file checks after a change to the FX kernels: GPU tests of those rows, the FX bench (chain + reverb), the
# normaliser per stem, the runner over two songs (cold and warm pass)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/fxq; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "normaliz or reverb or fx or cli or fir or chain or feature" 2>&1 | tail -4 > $O/pytest.log; cat $O/pytest.log
timeout 300 python tools/bench_fx.py > $O/bench_fx.json 2>$O/bench_fx.err; python -c "
import json; d=json.load(open('$O/bench_fx.json')); print('chain ms', d['ms_per_chain'], 'reverb', d['conv_reverb'])"
timeout 300 python tools/prof_normalizer_kernels.py 3 2>&1 | tail -2 | tee $O/nz_times.txt
python - > $O/f2f.json 2>$O/f2f.err <<'PY'
import sys, json, contextlib
sys.path.insert(0, "tools")
import bench_cli
with contextlib.redirect_stdout(sys.stderr):
    r = bench_cli.run(180.0, "bf16", songs=2)
print(json.dumps(r))
PY
python -c "
import json; d=json.load(open('$O/f2f.json')); print({k: d[k] for k in ('setup_s','first_pass_s_per_song','value')})"
python - <<'PY' 2>/dev/null | tail -3
import sys; sys.path.insert(0, "tools")
import bench_normalizer, json
r = bench_normalizer.run(180.0, 6.0, "drums,other")
print({k: r[k] for k in ("value", "s_per_stem", "rel_dev_vs_oracle_on_excerpt")})
PY

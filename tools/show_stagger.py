import json, sys, glob, numpy as np
for f in sorted(glob.glob('gpurun_out/stag_*.json')):
    tag = f.split('stag_')[1][:-5]
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(tag, round(d['value'], 1), 'seg/s', round(d['ms_per_step'], 2), 'ms  TCN', round(d['roofline']['achieved'], 1), 'TF', round(d['roofline']['avg_launch_ms'], 3), 'ms')
    except Exception as e:
        print(tag, 'ERR', e); continue
    try:
        a = np.fromfile(f'gpurun_out/stag_{tag}.bin', dtype=np.int64).reshape(-1, 10)
        a = a[a[:, 0] != 0]
        seq = a[:, [0, 1, 2, 6, 7, 3]]
        d = np.diff(seq, axis=1)
        idx = np.argsort(a[:, 0])
        late = idx[len(idx) // 2:]
        print('   stage/main/xin/epi/store (2nd half of tiles):', d[late].mean(0).round(0), 'total', (seq[late, -1] - seq[late, 0]).mean().round(0),
              ' kernel span', (a[:, 3].max() - a[:, 0].min()))
    except Exception as e:
        print('   no phases', e)

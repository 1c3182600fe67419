import json, sys, glob, numpy as np
for f in sorted(glob.glob('gpurun_out/duo_*_*.json')):
    tag = f.split('duo_')[1][:-5]
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(tag, round(d['value'], 1), 'seg/s', round(d['ms_per_step'], 2), 'ms  TCN', round(d['roofline']['achieved'], 1), 'TF', [round(x, 2) for x in d['roofline']['per_block_ms'][1:14]])
    except Exception as e:
        print(tag, 'ERR', e); continue
    if tag.startswith('0_'):
        continue
    try:
        a = np.fromfile(f'gpurun_out/duo_phase_{tag}.bin', dtype=np.int64).reshape(-1, 10)
        for s in (0, 1):
            b = a[s::2]; b = b[b[:, 0] != 0]
            print('   set', s, 'main %.0f | barrier %.0f | epilogue(+prefetch) %.0f | stage store %.0f | total %.0f' % ((b[:, 2] - b[:, 0]).mean(), (b[:, 6] - b[:, 2]).mean(),
                  (b[:, 7] - b[:, 6]).mean(), (b[:, 3] - b[:, 7]).mean(), (b[:, 3] - b[:, 0]).mean()))
    except Exception as e:
        print('   no phases', e)

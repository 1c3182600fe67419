import json, sys, numpy as np
for v in sys.argv[1:]:
    try:
        d = json.load(open(f'gpurun_out/var_{v}.json'))
        print(v, round(d['value'], 1), 'seg/s', round(d['ms_per_step'], 2), 'ms/step  TCN', round(d['roofline']['achieved'], 1), 'TF',
              [round(x, 2) for x in d['roofline']['per_block_ms']])
    except Exception as e:
        print(v, 'ERR', e, open(f'gpurun_out/var_{v}.err').read()[-400:])
    try:
        a = np.fromfile(f'gpurun_out/phase_v{v}.bin', dtype=np.int64).reshape(-1, 4)
        d = np.diff(a, axis=1)
        print('   phases (cycles): stage %.0f main %.0f epi %.0f total %.0f' % (d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), (a[:, 3] - a[:, 0]).mean()))
    except Exception as e:
        print('   no phase file', e)

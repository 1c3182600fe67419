mkdir -p gpurun_out/r04_probe
( which amd-smi rocm-smi; amd-smi version; amd-smi metric -g 0 --clock --power --json 2>&1 | head -80; rocm-smi --showclocks --showpower 2>&1 | head -40; ls /sys/class/drm/card*/device/pp_dpm_sclk; cat /sys/class/drm/card*/device/pp_dpm_sclk; ls /sys/class/drm/card*/device/hwmon/*/; cat /sys/class/drm/card*/device/hwmon/*/power1_average /sys/class/drm/card*/device/hwmon/*/freq1_input 2>&1 ) > gpurun_out/r04_probe/smi.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_probe/bench_driver_cmd.json 2> gpurun_out/r04_probe/bench_driver_cmd.err
tail -c 600 gpurun_out/r04_probe/bench_driver_cmd.json

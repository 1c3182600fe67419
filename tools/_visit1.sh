cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
./tools/micro/fft_bin_gemm > gpurun_out/r04_micro_fft_bin_gemm.txt 2>&1
cat gpurun_out/r04_micro_fft_bin_gemm.txt
bash tools/gpu_fx_pmc.sh

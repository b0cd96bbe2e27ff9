mkdir -p gpurun_out/final && export TMPDIR=/tmp && \
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final/gpu_tests.txt 2>&1 && tail -2 gpurun_out/final/gpu_tests.txt && \
timeout -k 10 300 python tools/gpu_validate_config3.py 400 > gpurun_out/final/validate_config3.json 2> gpurun_out/final/validate_config3.err && \
timeout -k 10 300 python tools/gpu_validate_config2.py 1500 > gpurun_out/final/validate_config2.json 2> gpurun_out/final/validate_config2.err && \
bash tools/profile_bench.sh r04_config3 k_joint --config 3 && echo prof3 ok && \
bash tools/profile_bench.sh r04_config5 k_sweep_ --config 5 && echo prof5 ok && \
bash tools/profile_bench.sh r04 k_sweep_ && echo prof2 ok

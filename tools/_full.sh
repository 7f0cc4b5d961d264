timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/t_full.txt 2>&1
tail -3 gpurun_out/t_full.txt
timeout 600 python bench.py --full-record gpurun_out/r06_bench7_full.json > gpurun_out/r06_bench7_line.json 2> gpurun_out/r06_bench7.err
tail -c 600 gpurun_out/r06_bench7_line.json

python -m pytest tests/test_gpu_band_direct.py tests/test_gpu_parity.py tests/test_viewgraph.py -x -q -m gpu 2>&1 | grep -a "passed\|failed" | tail -2
for e in A=1 IROTAVG_NO_LAST_GUESS=1 A=1 IROTAVG_NO_LAST_GUESS=1; do env $e python bench.py --no-pmc --no-kernels --no-cpu --no-extra > gpurun_out/ab.json 2>/dev/null; python tools/print_line.py gpurun_out/ab.json; done

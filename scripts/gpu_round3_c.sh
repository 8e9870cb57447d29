#!/bin/bash
mkdir -p gpurun_out/r3c
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3c/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_envelopes.py -x -q -m gpu > gpurun_out/r3c/t2.log 2>&1; echo "t2 rc $?" >> gpurun_out/r3c/rc.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err; echo "bench rc $?" >> gpurun_out/r3c/rc.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3c/all.log 2>&1; echo "all rc $?" >> gpurun_out/r3c/rc.log
tail -n 3 gpurun_out/r3c/t2.log gpurun_out/r3c/all.log; cat gpurun_out/r3c/rc.log

#!/bin/bash
# first GPU call of round 3: the new parity tests, the guard sweep, a short bench
mkdir -p gpurun_out/r3a
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r3a/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_config3.py tests/test_gpu_longtarget.py -x -q -m gpu > gpurun_out/r3a/t1.log 2>&1; echo "t1 rc $?" >> gpurun_out/r3a/rc.log
timeout 600 python scripts/oa_guard_sweep.py > gpurun_out/r3a/sweep.log 2>&1; echo "sweep rc $?" >> gpurun_out/r3a/rc.log
timeout 900 python -m pytest tests/test_gpu_envelopes.py -x -q -m gpu > gpurun_out/r3a/t2.log 2>&1; echo "t2 rc $?" >> gpurun_out/r3a/rc.log
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc $?" >> gpurun_out/r3a/rc.log
tail -3 gpurun_out/r3a/*.log

#!/bin/bash
# run the GPU test suite repeatedly; if a run does not exit within 75 s, dump the native stacks of the stuck process
for i in 1 2 3 4 5 6; do
  python -X faulthandler -m pytest tests -x -q -m gpu --timeout 60 > /tmp/pt.$i 2>&1 &
  pid=$!
  for t in $(seq 1 75); do sleep 1; kill -0 $pid 2>/dev/null || break; done
  if kill -0 $pid 2>/dev/null; then
    echo "run $i: STUCK after 75 s; last output:"; tail -3 /tmp/pt.$i
    if command -v gdb >/dev/null; then gdb -p $pid -batch -ex "thread apply all bt 12" 2>/dev/null | grep -v "^\[New\|^\[Thread" | head -150; else echo "no gdb"; cat /proc/$pid/status | head -5; for t in /proc/$pid/task/*; do echo "$t $(cat $t/comm) $(cat $t/wchan 2>/dev/null)"; done | head -60; fi
    kill -9 $pid; exit 0
  fi
  wait $pid; echo "run $i: rc=$? $(tail -1 /tmp/pt.$i)"
done

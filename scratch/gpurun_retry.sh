#!/bin/bash
# gpurun with retries while the pod has no free GPU slot (exit code 3: nothing charged)
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3

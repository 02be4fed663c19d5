#!/bin/bash
# usage: scripts/_grun.sh NAME TIMEOUT -- retries `gpurun -- bash scripts/_job_NAME.sh` while no slot is free (rc 3)
name=$1; to=${2:-900}
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $to -- bash scripts/_job_$name.sh > gpurun_out/${name}_call.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "done rc=$rc" >> gpurun_out/${name}_call.log; exit $rc; fi
  sleep 45
done

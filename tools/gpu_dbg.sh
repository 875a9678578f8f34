#!/bin/bash
mkdir -p gpurun_out/dbg
timeout 300 python bench.py --group --no-cpu-baseline --no-api --steps 5 > gpurun_out/dbg/group.out 2> gpurun_out/dbg/group.err
echo "rc=$?" >> gpurun_out/dbg/group.err
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 timeout 300 python tests/dist_api_script.py > gpurun_out/dbg/api.out 2> gpurun_out/dbg/api.err
echo "rc=$?" >> gpurun_out/dbg/api.err
tail -n 30 gpurun_out/dbg/*.err

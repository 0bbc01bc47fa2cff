#!/bin/bash
mkdir -p gpurun_out
timeout 600 python scratch/mb_rollout_batch.py > gpurun_out/r02_rollout_batch.txt 2>&1; echo rc=$?
tail -3 gpurun_out/r02_rollout_batch.txt

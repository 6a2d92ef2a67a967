#!/bin/bash
# r05t: the step's torch glue launches by source line (torch.profiler stacks)
mkdir -p gpurun_out
timeout 400 python profiles/glue_trace.py > gpurun_out/r05t_glue_trace.txt 2>&1
head -70 gpurun_out/r05t_glue_trace.txt

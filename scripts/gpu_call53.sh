#!/bin/bash
timeout 300 python scripts/bf16_small_time.py 2>&1 | grep fwd | tee gpurun_out/c53_small.txt

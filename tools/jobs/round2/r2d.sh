#!/bin/bash
# round 2, GPU call D: re-run of the tests fixed after call C + examples
O=gpurun_out/r2d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt
tail -n 30 $O/pytest_all.log | cut -c1-250

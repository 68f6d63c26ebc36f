cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_frame_submission.py -m gpu -x -q -k two_sessions > gpurun_out/r6/c5_tests.txt 2>&1; grep -n "thread [01]:" gpurun_out/r6/c5_tests.txt | head; tail -3 gpurun_out/r6/c5_tests.txt

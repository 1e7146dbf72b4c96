#!/bin/bash
# Why does capture-after-eager-backward abort?  Runs tools/probes/graph_recapture.py eager_then_capture with HIP logging and keeps the tail.
cd $GRAFT_REPO_ROOT
AMD_LOG_LEVEL=3 timeout -k 5 120 python tools/probes/graph_recapture.py eager_then_capture > gpurun_out/recapture_stdout.txt 2> gpurun_out/recapture_hiplog.txt
echo "exit code $?"
tail -5 gpurun_out/recapture_stdout.txt
grep -n "hipStreamEndCapture\|hipErrorStreamCapture\|capture\|Capture" gpurun_out/recapture_hiplog.txt | tail -40
tail -30 gpurun_out/recapture_hiplog.txt | cut -c1-300

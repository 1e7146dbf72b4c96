set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fcal -- python3 $R/tools/probes/fetch_calibration.py > $O/fetch_calibration_log.txt 2>&1
python3 $R/tools/pmc_sq.py /tmp/fcal spconv_reduce > $O/fetch_calibration.txt; cat $O/fetch_calibration_log.txt | grep -v Warn | tail -6; cat $O/fetch_calibration.txt | cut -c1-200

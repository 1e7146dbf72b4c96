# SQ counters of the attention kernels at the ViT's shape (batch 4 and 1), one launch of each tiling
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/attn_sq -- python3 $R/tools/bench_attn.py --batches 4 --iters 1 > /dev/null 2>&1
python3 $R/tools/pmc_sq.py /tmp/attn_sq attn_ > $O/attn_pmc_sq.txt
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/attn_sq2 -- python3 $R/tools/bench_attn.py --batches 4 --iters 1 > /dev/null 2>&1
python3 $R/tools/pmc_sq.py /tmp/attn_sq2 attn_ > $O/attn_pmc_sq2.txt
cat $O/attn_pmc_sq.txt | cut -c1-220 | head -60

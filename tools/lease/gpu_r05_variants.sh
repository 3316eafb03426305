#!/bin/bash
# read-kernel variants side by side (RSQ_SPEC_OPTIONS, rsq_spec.h): bench line + VALU counters each.  bash tools/lease/gpu_r05_variants.sh "<opts 1>" "<opts 2>" ...   ("-" = none)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then unset RSQ_SPEC_OPTIONS; else export RSQ_SPEC_OPTIONS="$v"; fi
  echo "=== variant $i: $v"
  bash tools/gpu_quick.sh r05_var_$i 2>&1 | grep -v "^ *\"SQ_INSTS_SALU\|amdgpu.ids" | grep "BENCH\|SQ_INSTS_VALU\|SQ_ACTIVE_INST_VALU\|valu_busy\|kernel_ms\|no bench"
done

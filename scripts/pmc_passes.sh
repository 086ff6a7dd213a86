#!/bin/bash
# Two rocprofv3 --pmc passes (own runs, --kernel-trace only) of one command, then the per-kernel summary of scripts/pmc_kernel_counters.py:
#   pass A: six SQ counters + GRBM_GUI_ACTIVE (busy fractions, stalls);   pass B: the LDS pair + GRBM_GUI_ACTIVE (bank conflicts) — its own pass: gfx950 has eight
#   SQ slots, and a ninth counter in one pass comes back stale.
# usage: scripts/pmc_passes.sh "<label>" "<kernel name substring>[;<second substring>...]" <command ...>
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
label="$1"; pats="$2"; shift 2
A="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
rm -rf /tmp/pmc_a /tmp/pmc_b
rocprofv3 --kernel-trace --pmc $A --output-format csv -d /tmp/pmc_a -o p -- "$@" > /tmp/pmc_a.log 2>&1 || tail -3 /tmp/pmc_a.log
rocprofv3 --kernel-trace --pmc $B --output-format csv -d /tmp/pmc_b -o p -- "$@" > /tmp/pmc_b.log 2>&1 || tail -3 /tmp/pmc_b.log
IFS=';' read -ra PP <<< "$pats"
for pat in "${PP[@]}"; do
  python scripts/pmc_kernel_counters.py /tmp/pmc_a,/tmp/pmc_b "$pat" "$label [$pat]"
done

#!/bin/bash
# GPU box: which engine carries the downloads, per submission pattern (see copy_engine_probe.hip)
hipcc --offload-arch=gfx950 -O3 ${GRAFT_REPO_ROOT:-/root/repo}/scripts/copy_engine_probe.hip -o /tmp/cep 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3 4 5; do
  rm -rf /tmp/cep_out
  rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/cep_out -- /tmp/cep $v > /dev/null 2>&1
  d2h=$(grep -c DEVICE_TO_HOST /tmp/cep_out/*/*_memory_copy_trace.csv 2>/dev/null)
  blit=$(grep -c copyBuffer /tmp/cep_out/*/*_kernel_trace.csv 2>/dev/null)
  echo "variant $v: DMA-engine downloads $d2h, copy-kernel launches $blit"
done

#!/bin/bash
# Can the one-GPU lease be split into two logical devices?  READ-ONLY probe: what the box shows about compute /
# memory partitioning and whether the switch is reachable at all.  Nothing is written to the driver -- a partition switch re-creates
# the GPU's KFD nodes, and a box returned in another mode (or a switch that fails half-way) would take the GPU away from whoever
# gets the box next.   tools/partition_probe.sh > gpurun_out/r06/partition_probe.log
echo "== id / container"; id; cat /proc/1/cgroup 2>/dev/null | head -3; grep -E ' /sys | /sys/' /proc/mounts | head -5
echo "== devices"; ls -la /dev/kfd /dev/dri/ 2>&1
echo "== rocm-smi partitions"; rocm-smi --showcomputepartition --showmemorypartition 2>&1 | head -30
echo "== amd-smi"; (amd-smi partition 2>&1 || amd-smi static --partition 2>&1) | head -60
echo "== sysfs"
for c in /sys/class/drm/card*/device; do
  [ -e $c/current_compute_partition ] || continue
  echo "$c: current_compute_partition=$(cat $c/current_compute_partition 2>&1) available=$(cat $c/available_compute_partition 2>&1)"
  echo "   current_memory_partition=$(cat $c/current_memory_partition 2>&1) available=$(cat $c/available_memory_partition 2>&1)"
  ls -la $c/current_compute_partition; [ -w $c/current_compute_partition ] && echo "   writable by this user: yes" || echo "   writable by this user: no"
done
echo "== KFD topology nodes"; for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "$n simd_count=$(grep -E '^simd_count' $n/properties 2>/dev/null | cut -d' ' -f2) gfx=$(grep -E '^gfx_target_version' $n/properties 2>/dev/null | cut -d' ' -f2) xcc=$(grep -E '^num_xcc' $n/properties 2>/dev/null | cut -d' ' -f2)"; done
echo "== HIP view"
python - <<'PY'
import torch
print("hipGetDeviceCount", torch.cuda.device_count())
for i in range(torch.cuda.device_count()):
    p = torch.cuda.get_device_properties(i)
    print(i, p.name, "CUs", p.multi_processor_count, "mem GiB", round(p.total_memory / 2**30, 1))
PY
echo "== env"; env | grep -E 'HIP_VISIBLE|ROCR_VISIBLE|CUDA_VISIBLE|GPU_DEVICE_ORDINAL|HSA_' 
echo "== rocminfo agents"; rocminfo 2>/dev/null | grep -E 'Marketing Name|Compute Unit|Uuid|Node:' | head -20

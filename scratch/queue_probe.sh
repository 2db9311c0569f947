#!/bin/bash
cd "$(dirname "$0")/.."
for s in plain pg pg_barrier hi1 hi2 hi3 hi4 norm1 norm2 norm3 low1 low2 plain; do
  timeout 300 python scratch/queue_probe.py $s 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
done

#!/bin/bash
# Same-box A/B of the layered kernels' occupancy (3 waves per SIMD with spills, shipped, against 2
# without): first  tools/build_variant.sh layered2 "-DXRT_LAYERED_WAVES=2" reflect_layered_f reflect_layered_x
#   gpurun -- 'bash tools/ab_layered_waves.sh'   -> profiles/r05_layered_waves_ab.txt
echo "== layered kernels: 3 waves per SIMD (168 VGPRs, spills; shipped)"; XRT_HIP_NO_FUSE=1 python tools/probe_multilayer.py 2>&1 | grep "ms /"
echo "== layered kernels: 2 waves per SIMD (-DXRT_LAYERED_WAVES=2: 256 VGPRs)"; XRT_HIP_NO_FUSE=1 XRT_HIP_LIBRARY=$PWD/xrt_amd/ab/libxrt_layered2.so python tools/probe_multilayer.py 2>&1 | grep "ms /"
python - <<'PY'
import sys
sys.path.insert(0,'tools')
import kernel_resources as kr
for lib in ('xrt_amd/libxrt_hip.so','xrt_amd/ab/libxrt_layered2.so'):
    t = kr.kernels(lib)
    for k,v in t.items():
        if 'reflect_fusedINS_4SpecILi0ELin1ELi5ELb0EEELi0' in k or 'reflect_fused_xtalINS_4SpecILi0ELin1ELi5ELb0EEELi0' in k:
            print(lib, k[:70], v)
PY

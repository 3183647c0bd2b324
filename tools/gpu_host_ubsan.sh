#!/bin/bash
# Host-side UBSan + libstdc++ container assertions over the GPU suite: the extension's host code (graph/session
# planners, CSR builds, heavy-row splits, arenas, the CDCL filter) built with
#   DSAT_CXXFLAGS="-Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined -D_GLIBCXX_ASSERTIONS -g" \
#     python -c "from diffusionsat_amd import build; build.build(out='tools/libs/libdsat_ubsan.so')"
# (device code is not instrumented).  AddressSanitizer is not usable with a GPU in the process on this pool: ROCm's
# ASan runtime intercepts hsa_amd_memory_pool_allocate and needs XNACK; the CPU-only tests do run under it
# (LD_PRELOAD=libclang_rt.asan-x86_64.so python -m pytest tests/test_host_sat.py tests/test_cabi_cpu.py).
set -o pipefail
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.ubsan_standalone-x86_64.so
export DSAT_LIB=$PWD/tools/libs/libdsat_ubsan.so
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
mkdir -p gpurun_out
LD_PRELOAD=$RT timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/host_ubsan.log 2>&1
rc=$?
tail -15 gpurun_out/host_ubsan.log
exit $rc

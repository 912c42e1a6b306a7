#!/bin/bash
# developer check (CPU, no GPU needed): the host side of libnam_hip.so (loader, planner, JIT driver, ABI) rebuilt with
# AddressSanitizer + UBSan, then the CPU test suite and the planner fuzz run against it. The shipped library is put back.
set -e
cd "$(dirname "$0")/.."
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
mkdir -p variants/obj_asan
for f in nam_hip_api api_launch api_session api_host_io nam_loader plan plan_ops plan_a1 plan_wr wr_jit; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fvisibility=hidden -fsanitize=address,undefined -fno-gpu-sanitize \
    -fno-omit-frame-pointer -x hip -c -o variants/obj_asan/$f.o neuralampmodelercore_amd/csrc/$f.cpp &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=address,undefined -shared-libsan -o variants/libnam_hip_asan.so \
  neuralampmodelercore_amd/lib/obj/kernel_*.o variants/obj_asan/*.o
LIB=neuralampmodelercore_amd/lib/libnam_hip.so
cp $LIB variants/libnam_hip_shipped.so
trap 'cp variants/libnam_hip_shipped.so $LIB; rm -rf variants/libnam_hip_shipped.so variants/libnam_hip_asan.so variants/obj_asan' EXIT
cp variants/libnam_hip_asan.so $LIB
rm -f variants/asan_log.*
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:log_path=$PWD/variants/asan_log UBSAN_OPTIONS=print_stacktrace=1:log_path=$PWD/variants/asan_log
# (the compiler-crash test makes clang segfault ON PURPOSE: with the sanitizer runtime preloaded into the child too, that crash may be
# reported as one of ours)
python -m pytest tests/ -x -q -m "not gpu" -p no:cacheprovider -k "not survives_a_compiler_crash" | tail -15
python tools/fuzz_models.py 160 12 --load-only | tail -1
unset LD_PRELOAD
if ls variants/asan_log.* > /dev/null 2>&1; then echo "sanitizer reports:"; cat variants/asan_log.* | head -60; exit 1; fi
echo "host side clean under ASan + UBSan"

#!/bin/bash
# tools/asan_host_check.sh build | run   [address | thread | undefined | control]      (default: address)
# `control` = the thread-sanitizer build with -DTFHE_TSAN_CONTROL (one deliberately unsynchronised counter in combine_request): `run control`
# SUCCEEDS only if ThreadSanitizer reports that race -- the positive control for "0 reports" of `run thread`.
# The product's HOST code (csrc/tfhe_hip.hip: staging, grow-only buffers, the combiner's request queue, clone_to, key blobs) under
# AddressSanitizer; the device code is not instrumented (-fno-gpu-sanitize), the blind-rotate units are the shipped objects' sources
# at their normal flags.  `build` cross-compiles here (no GPU needed) into go-tfhe_amd/lib/variants/asan/ (git-ignored, travels to the
# GPU box); `run` executes, on a GPU box, the C++ host-mirror test and the concurrent-submitter bench against that library.
set -e
cd "$(dirname "$0")/.."
SAN=${2:-address}
D=go-tfhe_amd/lib/variants/${SAN}san
CONTROL=""; if [ $SAN = control ]; then SAN=thread; CONTROL=-DTFHE_TSAN_CONTROL; fi
GPUSAN=""; [ $SAN = address ] && GPUSAN=-fno-gpu-sanitize
RTNAME=asan; [ $SAN = thread ] && RTNAME=tsan; [ $SAN = undefined ] && RTNAME=ubsan_standalone
HIPCC=/opt/rocm/bin/hipcc
CLANG=/opt/rocm/lib/llvm/bin/clang++
if [ "$1" = build ]; then
    O=/tmp/${SAN}san_obj; mkdir -p $D $O
    F="--offload-arch=gfx950 -std=c++17 -fPIC"
    ILP="-mllvm -amdgpu-sched-strategy=max-ilp"
    $HIPCC $F -O1 -g -fsanitize=$SAN $GPUSAN $CONTROL -shared-libsan -Wno-option-ignored -c go-tfhe_amd/csrc/tfhe_hip.hip -o $O/a.o &
    $HIPCC $F -O3 $ILP -c go-tfhe_amd/csrc/blind_rotate.hip -o $O/b.o 2>/dev/null &
    $HIPCC $F -O3 $ILP -mllvm -enable-post-misched=0 -c go-tfhe_amd/csrc/blind_rotate_oct.hip -o $O/c.o 2>/dev/null &
    $HIPCC $F -O3 -c go-tfhe_amd/csrc/blind_rotate_n2048.hip -o $O/d.o 2>/dev/null &
    wait
    $HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=$SAN $GPUSAN -shared-libsan -Wno-option-ignored $O/{a,b,c,d}.o -o $D/libtfhe_hip.so
    RT=$(dirname $($CLANG -print-file-name=libclang_rt.$RTNAME-x86_64.so))
    for t in tests/cpp/test_host_mirror tools/combine_bench; do
        $CLANG -O1 -g -std=c++17 -fsanitize=$SAN -shared-libsan $t.cpp -o $D/$(basename $t) -L$D -ltfhe_hip -Loracle -ltfhe_oracle \
            -L/opt/rocm/lib -lpthread -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../../../oracle' -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$RT
    done
    ls -la $D
else
    # detect_leaks=0: the HIP runtime keeps process-lifetime allocations; protect_shadow_gap=0: the ROCm runtime maps into the gap
    export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1
    # thread: the HIP / HSA runtimes are not instrumented; reports whose frames all lie inside them are theirs (suppressed by library
    # name), every report that touches libtfhe_hip.so's own frames counts
    printf 'called_from_lib:libamdhip64.so\ncalled_from_lib:libhsa-runtime64.so\nrace:libamdhip64.so\nrace:libhsa-runtime64.so\n' > /tmp/tsan.supp
    export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
    export TSAN_OPTIONS=suppressions=/tmp/tsan.supp:halt_on_error=0:second_deadlock_stack=1:exitcode=66:ignore_noninstrumented_modules=0
    # The ROCm ASan runtime owns a device allocator whose teardown check ("dev_runtime_unloaded_") can fire inside libamdhip64's own
    # exit-time finaliser (__cxa_finalize -> libhsa-runtime64 -> operator delete), after main has returned: that report is about the
    # runtime's unload order, not about this library -- it is recognised by its text and its frames and tolerated; anything else fails.
    run() {
        echo "+ $*"
        "$@" > /tmp/asan_out.txt 2>&1 && { cat /tmp/asan_out.txt; return 0; }
        cat /tmp/asan_out.txt | grep -v "^    #"
        if [ $SAN = thread ]; then echo "TSAN RUN REPORTED: $*"; FAILED=1; return 0; fi
        if grep -q "dev_runtime_unloaded_" /tmp/asan_out.txt && grep -q "__cxa_finalize" /tmp/asan_out.txt \
           && ! grep -q "ERROR: AddressSanitizer" /tmp/asan_out.txt; then
            echo "(exit-time CHECK of the ROCm ASan runtime inside the HIP runtime's finaliser: tolerated)"; return 0
        fi
        echo "ASAN RUN FAILED: $*"; exit 1
    }
    if [ -n "$CONTROL" ]; then
        echo "+ $D/combine_bench 64   (control build: a report is EXPECTED)"
        $D/combine_bench 64 > /tmp/asan_out.txt 2>&1 || true
        grep -v "^    #" /tmp/asan_out.txt | grep -A 12 "WARNING: ThreadSanitizer: data race" | head -30
        if grep -q "WARNING: ThreadSanitizer: data race" /tmp/asan_out.txt && grep -q "g_tsan_control" /tmp/asan_out.txt; then
            echo "control: ThreadSanitizer reports the seeded race in combine_request -- the instrumentation sees this code"; exit 0
        fi
        echo "CONTROL FAILED: the seeded race was not reported"; exit 1
    fi
    run $D/test_host_mirror
    run $D/combine_bench 64
    run $D/combine_bench 256
    run $D/combine_bench 64 pbs
    run $D/combine_bench 4 batch 700 3          # mid-size host batches from several threads: the overlapped upload / kernels / download path
    run $D/combine_bench 3 batch 300 2 pbs      # ... and of programmable bootstraps
    [ -n "$FAILED" ] && exit 1
    echo "host code under ${SAN} sanitizer: no report"
fi

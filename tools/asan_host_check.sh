#!/bin/bash
# Host-side AddressSanitizer check of the library (device code is not
# instrumented): builds asan_build/libslamfe.so and the C++ host-adapter test
# against it.  Build here (no GPU needed), then run `tools/asan_host_check.sh run`
# on a GPU box.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=$ROOT/asan_build
CL=/opt/rocm/lib/llvm/bin/clang
if [ "$1" != "run" ]; then
    mkdir -p "$B"
    cd "$ROOT/slam_amd/csrc"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function \
        -mllvm -amdgpu-mfma-vgpr-form -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer \
        -I../../include -I. -shared -o "$B/libslamfe.so" slamfe_api.hip slamfe_k_image.hip slamfe_k_select.hip \
        slamfe_k_describe.hip slamfe_k_match.hip slamfe_k_map.hip
    ${CL}++ -O1 -g -std=c++14 -fsanitize=address -fno-omit-frame-pointer -I"$ROOT/include" -I"$ROOT/slam_amd/host" \
        -I"$ROOT/oracle" "$ROOT/tests/cpp/test_host_adapter.cpp" -o "$B/test_host_adapter" \
        -L"$B" -lslamfe -L"$ROOT/oracle" -lslamfe_oracle -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../oracle' \
        -Wl,-rpath,/opt/rocm/lib
    ${CL}++ -O1 -g -std=c++14 -fsanitize=address -fno-omit-frame-pointer -I"$ROOT/include" -I"$ROOT/slam_amd/host" \
        -I"$ROOT/oracle" "$ROOT/tests/cpp/test_tracking_loop.cpp" -o "$B/test_tracking_loop" \
        -L"$B" -lslamfe -L"$ROOT/oracle" -lslamfe_oracle -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../oracle' \
        -Wl,-rpath,/opt/rocm/lib
    $CL -x c -O1 -g -std=c99 -fsanitize=address -fno-omit-frame-pointer -I"$ROOT/include" \
        "$ROOT/tests/cpp/abi_batch_check.c" -o "$B/abi_batch_check" -L"$B" -lslamfe -Wl,-rpath,'$ORIGIN' \
        -Wl,-rpath,/opt/rocm/lib
    echo built
    exit 0
fi
cd "$ROOT"
python - <<'PY'
from slam_amd import synth
a, b = synth.frame_pair(640, 480)
a.tofile("asan_build/a.raw"); b.tofile("asan_build/b.raw")
synth.frame_stream(320, 240, 40, seed=15).tofile("asan_build/stream.raw")
PY
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0
"$B/test_host_adapter" asan_build/a.raw asan_build/b.raw 640 480
"$B/abi_batch_check"
"$B/test_tracking_loop" asan_build/stream.raw 320 240 40 3 2

#!/bin/bash
# Address + undefined-behaviour sanitizer pass over the HOST side (the device code cannot be sanitized on this pool): the host library
# built with g++ -fsanitize=address,undefined and the CPU tests that drive it run against that build; the native CPU tests built as one
# sanitized binary with clang. Optional: `tools/sanitize_host.sh thread` runs the parallel BVH build under the thread sanitizer.
# Outputs under /tmp/hipr_sanitize. Found so far: a use-after-free in the scene managers (a name passed by reference into the storage
# create() grows), fixed by taking names by value.
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/hipr_sanitize
mkdir -p $out
cd $root/bifrost3d_amd
SRCS=$(sed -n 's/^HOST_SRCS := //p' Makefile)
mode=${1:-address}
if [ "$mode" = thread ]; then
    g++ -O1 -g -std=c++17 -fPIC -fsanitize=thread -shared -o $out/libhiprenderer_host.so $SRCS -Lcsrc -lhiprenderer -lz -Wl,-rpath,$PWD/csrc || exit 1
    cat > $out/run.py <<PY
import sys
from pathlib import Path
sys.path.insert(0, '$root')
from bifrost3d_amd import capi
capi.HOST_LIB_PATH = Path('$out/libhiprenderer_host.so')
sys.argv = ['bvh_build_probe.py', '600000']
exec(open('$root/tools/bvh_build_probe.py').read())
PY
    LD_PRELOAD=$(g++ -print-file-name=libtsan.so) HIPR_BVH_THREADS=8 TSAN_OPTIONS="report_signal_unsafe=0" python $out/run.py > $out/thread.txt 2>&1
    echo "thread sanitizer: $(grep -c 'WARNING: ThreadSanitizer' $out/thread.txt) reports, see $out/thread.txt"; tail -2 $out/thread.txt
    exit 0
fi
g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $out/libhiprenderer_host.so $SRCS -Lcsrc -lhiprenderer -lz -Wl,-rpath,$PWD/csrc || exit 1
cat > $out/run.py <<PY
import sys
from pathlib import Path
sys.path.insert(0, '$root'); sys.path.insert(0, '$root/tests')
from bifrost3d_amd import capi
capi.HOST_LIB_PATH = Path('$out/libhiprenderer_host.so')
import pytest
sys.exit(pytest.main(['-q', '-m', 'not gpu', 'tests/test_image_codecs_cpu.py', 'tests/test_loaders_cpu.py', 'tests/test_coverage_cpu.py', 'tests/test_host_cpu.py', '-k', 'not parallel_bvh', '-p', 'no:cacheprovider']))
PY
cd $root
LD_PRELOAD="$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 python $out/run.py > $out/python_tests.txt 2>&1
echo "python CPU tests on the sanitized host library: $(tail -1 $out/python_tests.txt); $(grep -c 'runtime error\|AddressSanitizer' $out/python_tests.txt) sanitizer reports"
cd $root/bifrost3d_amd
TESTS=$(ls ../tests/native/*.cpp)
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -o $out/renderer_test $TESTS $SRCS \
    -Lcsrc -lhiprenderer -L/opt/rocm/lib -lamdhip64 -lz -lpthread -Wl,-rpath,$PWD/csrc -Wl,-rpath,/opt/rocm/lib || exit 1
cd $root
ASAN_OPTIONS=detect_leaks=0 $out/renderer_test --cpu > $out/native_tests.txt 2>&1
echo "native CPU tests, sanitized: $(tail -1 $out/native_tests.txt); $(grep -c 'runtime error\|AddressSanitizer' $out/native_tests.txt) sanitizer reports"

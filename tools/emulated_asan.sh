# The CPU-emulated kernel tests (tests/simt/) under AddressSanitizer: numpy's and torch's buffers are malloc'ed, so a
# kernel's read or write past the end of an argument buffer or workspace -- forgiven on a GPU while the page is mapped --
# is reported with the source line of the .hip file.  ~12 minutes on 16 cores.  (Not part of the pytest run: it needs
# the sanitizer runtime preloaded into the interpreter.)
#   bash tools/emulated_asan.sh [pytest arguments]
set -u
cd "$(dirname "$0")/.."
export SIMT_ASAN=1
export LD_PRELOAD=$(gcc -print-file-name=libasan.so)
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0
exec python -m pytest tests -q -m "not gpu" -k emulated "$@"

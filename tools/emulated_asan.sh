# The CPU-emulated kernel tests (tests/simt/) under AddressSanitizer: numpy's and torch's buffers are malloc'ed, so a
# kernel's read or write past the end of an argument buffer or workspace -- forgiven on a GPU while the page is mapped --
# is reported with the source line of the .hip file.  ~12 minutes on 16 cores.  (Not part of the pytest run: it needs
# the sanitizer runtime preloaded into the interpreter.)
#   bash tools/emulated_asan.sh [pytest arguments]
# SIMT_UBSAN=1 bash tools/emulated_asan.sh: UndefinedBehaviorSanitizer instead (shifts by the operand's width or more,
# misaligned vector accesses, signed overflow, out-of-range float -> int conversions); its reports go to /tmp/ubsan.log.*
set -u
cd "$(dirname "$0")/.."
export SIMT_BUILD_DIR=${SIMT_BUILD_DIR:-/tmp/simt_build_sanitizers}
if [ -n "${SIMT_UBSAN:-}" ]; then
  rm -f /tmp/ubsan.log.*
  export UBSAN_OPTIONS=print_stacktrace=0:log_path=/tmp/ubsan.log
  python -m pytest tests -q -m "not gpu" -k emulated "$@"
  echo "UndefinedBehaviorSanitizer reports: $(cat /tmp/ubsan.log.* 2>/dev/null | grep -c 'runtime error')"
  cat /tmp/ubsan.log.* 2>/dev/null | grep "runtime error" | sed 's/0x[0-9a-f]*/ADDR/g' | sort | uniq -c | sort -rn | head -40
  exit 0
fi
export SIMT_ASAN=1
export LD_PRELOAD=$(gcc -print-file-name=libasan.so)
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0
exec python -m pytest tests -q -m "not gpu" -k emulated "$@"

#!/bin/bash
# One command for the four "parity unpinned" notes of DESIGN.md section 2: name collation vs `samtools sort -n`, read depth
# vs `samtools depth -aa`, pileup counts vs `samtools mpileup -a`, and (with hisat2, the example_index and a clone of
# the reference) example/test00 + test01 end to end.  Runs only where the tools exist -- not on the GPU boxes of this
# pool, whose image has neither samtools nor hisat2.
#   bash tools/check_against_samtools.sh [--bam aligned.bam --index-prefix PREFIX] [--example /path/to/KIR_graph/example]
cd "$(dirname "$0")/.."
if ! command -v samtools > /dev/null; then
  echo "[SKIP] samtools is not installed here; nothing was compared"
  exit 0
fi
exec python3 tests/check_against_samtools.py "$@"

#!/bin/bash
# Stage the reference's own bench scripts (UNMODIFIED, byte for byte) into the git-ignored scratch directory _refstage/ so
# that ONE gpurun call can execute them on the MI355X through tools/run_reference_bench.py.  /root/reference
# does not exist on the GPU box; nothing under _refstage/ is ever committed (.gitignore) and `tools/stage_reference.sh clean`
# removes it again.  usage: tools/stage_reference.sh [clean]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=${LC_REFERENCE:-/root/reference}
DST=$ROOT/_refstage
if [ "$1" = clean ]; then rm -rf "$DST"; echo "removed $DST"; exit 0; fi
[ -d "$REF/kernels/hgemm" ] || { echo "no reference at $REF" >&2; exit 1; }
rm -rf "$DST"
mkdir -p "$DST/kernels/hgemm/tools" "$DST/kernels/flash-attn"
cp "$REF/kernels/hgemm/hgemm.py" "$DST/kernels/hgemm/"
cp "$REF/kernels/hgemm/tools/utils.py" "$DST/kernels/hgemm/tools/"
cp "$REF/kernels/flash-attn/flash_attn_mma.py" "$DST/kernels/flash-attn/"
( cd "$DST" && sha256sum kernels/hgemm/hgemm.py kernels/hgemm/tools/utils.py kernels/flash-attn/flash_attn_mma.py ) > "$DST/SHA256SUMS"
( cd "$REF" && sha256sum kernels/hgemm/hgemm.py kernels/hgemm/tools/utils.py kernels/flash-attn/flash_attn_mma.py ) | diff - "$DST/SHA256SUMS"
echo "staged (identical to $REF):"; cat "$DST/SHA256SUMS"

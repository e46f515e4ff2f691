#!/usr/bin/env bash
# Register budget of every kernel in the library as hipcc reports it (-Rpass-analysis=kernel-resource-usage): VGPRs, waves per SIMD,
# spills, LDS.  A tuned kernel that loses a wave per SIMD to an innocent-looking edit shows up here before it shows up in the
# bench (r05: one LDS-held index in the grouped probe scan, 7.0 -> 10.3 ms).  Needs no GPU.
#   tools/kernel_resources.sh [out.txt]      (default profiles/kernel_resources.txt); diff against the committed copy
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
out="${1:-$root/profiles/kernel_resources.txt}"
cd "$root/vecgo_amd/csrc"
tmp="$(mktemp -d)"
trap 'rm -rf "$tmp"' EXIT
for f in *.hip; do
    extra=""
    [ "$f" = "k_pq_train.hip" ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden \
        -I../../include $extra -c "$f" -o "$tmp/x.o" -Rpass-analysis=kernel-resource-usage 2>&1 |
        grep -E "Function Name|VGPRs:|AGPRs:|Occupancy|VGPRs Spill|LDS Size" |
        sed -E 's/.*remark: +//; s/ \[-Rpass-analysis=kernel-resource-usage\]//' |
        awk -v file="$f" '/^Function Name/ {if (name != "") print line; name=$3; line=file " " name; next} {gsub(/ +/, " "); line=line " | " $0} END {if (name != "") print line}' >> "$tmp/all.txt" &
    while [ "$(jobs -r | wc -l)" -ge 6 ]; do sleep 1; done
done
wait
filt="$(command -v c++filt || command -v llvm-cxxfilt || echo cat)"
sort "$tmp/all.txt" | while read -r file name rest; do echo "$file $(echo "$name" | "$filt" | cut -c1-120) $rest"; done > "$out"
echo "wrote $out ($(wc -l < "$out") kernels)"

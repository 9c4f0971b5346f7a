# usage: tools/kernel_regs.sh [lib.so] [grep-pattern]   — VGPR / AGPR / spills / code bytes of the simulation kernels in a built library (no GPU)
SO=$(readlink -f ${1:-$(dirname $0)/../tactilesimulation_amd/csrc/libtsim_hip.so}); PAT=${2:-k_forward|k_backward}
W=$(mktemp -d); cp $SO $W/lib.so; cd $W
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null 2>&1
for f in lib.so.*hipv4-amdgcn*; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f | grep -E "\.name:|\.vgpr_count|\.agpr_count|\.vgpr_spill_count|\.sgpr_spill_count" | paste - - - - - | sed 's/  */ /g;s/\.//g' > notes.txt
  /opt/rocm/lib/llvm/bin/llvm-readelf -s $f | awk '$4=="FUNC" {print $8, $3}' | sort -u > sizes.txt
  grep -E "$PAT" notes.txt | while read -r line; do
    n=$(echo "$line" | sed 's/.*name: \([^ ]*\).*/\1/'); sz=$(grep "^$n " sizes.txt | head -1 | awk '{print $2}')
    echo "$line code_bytes: $sz" | sed 's/- //'
  done
done
cd /; rm -rf $W

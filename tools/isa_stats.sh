# usage: tools/isa_stats.sh <lib.so> [kernel-mangled-substring]   — extract the gfx950 code object, print registers / spills and the
# static instruction mix of one kernel (default: k_forward<float, 8, false, 16>)
set -e
SO=$(readlink -f $1); K=${2:-_Z9k_forwardIfLi8ELb0ELi16ELb0EEv7FwdArgsIT_E}
W=$(mktemp -d); cd $W
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading $SO >/dev/null 2>&1 || true
CO=$(ls $(dirname $SO)/$(basename $SO).0.hipv4-amdgcn* 2>/dev/null || ls *hipv4-amdgcn*)
mv $CO co; rm -f $(dirname $SO)/$(basename $SO).0.host-* *host-x86* 2>/dev/null || true
/opt/rocm/lib/llvm/bin/llvm-readelf --notes co | grep -E "\.name:|\.vgpr_count|agpr_count|sgpr_spill|vgpr_spill|private_segment_fixed" | paste - - - - - - | grep "$K" | sed 's/  */ /g'
/opt/rocm/lib/llvm/bin/llvm-objdump -d co > all.s
a=$(grep -n "<$K>:" all.s | cut -d: -f1); b=$(awk -v a=$a 'NR>a && /^[0-9a-f]+ <.*>:$/ {print NR; exit}' all.s)
sed -n "${a},${b}p" all.s > k.s
echo "static instructions: $(grep -c '//' k.s)"
awk '{print $1}' k.s | grep -E "^[vsdg]" | sed 's/_e32$//;s/_e64$//;s/_dpp$/(dpp)/;s/_sdwa$//' | awk '{c[$1]++} END {v=s=d=g=0; for (k in c) { if (k ~ /^v_/) v+=c[k]; else if (k ~ /^s_/) s+=c[k]; else if (k ~ /^ds_/) d+=c[k]; else g+=c[k] } print "VALU",v,"SALU",s,"DS",d,"VMEM",g; print "readlane",c["v_readlane_b32"],"writelane",c["v_writelane_b32"],"accvgpr_r",c["v_accvgpr_read_b32"],"accvgpr_w",c["v_accvgpr_write_b32"],"v_mov",c["v_mov_b32"],"waitcnt",c["s_waitcnt"],"nop",c["s_nop"],"saveexec",c["s_and_saveexec_b64"],"cbranch_execz",c["s_cbranch_execz"]}'
cp k.s ${3:-/tmp/x/k_last.s}; rm -rf $W

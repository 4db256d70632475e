#!/bin/bash
# Development: compile one decoder instantiation for gfx950 and report the kernel's registers / scratch and where (between which
# barriers) the scratch instructions sit.   tools/dec_regs.sh [tu_dec_128_5] [extra -D flags]
TU=${1:-tu_dec_128_5}; shift
D=/tmp/dec_isa_$TU; mkdir -p $D; cd $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -Wno-pass-failed -fPIC "$@" -I/root/repo/efficientspeech_amd/csrc -c /root/repo/efficientspeech_amd/csrc/$TU.hip -o t.o -save-temps \
   -Rpass-analysis=kernel-resource-usage 2>&1 | tee log.txt | grep -A12 "Function Name: _ZN4esmi18mel_decoder_kernel" | grep -E "VGPRs:|Scratch|Spill|Occupancy"
grep -E " error" -A3 log.txt | head -20
S=$(ls *gfx950.s)
awk '/^_ZN4esmi18mel_decoder_kernel.*: / {on=1} on && /\.Lfunc_end/ {on=0} on {
   if ($1=="s_barrier") {printf("barrier %d: scratch st %d ld %d | ds_read %d ds_write %d mfma %d valu %d\n", nb++, st, ld, dr, dw, mf, va); st=ld=dr=dw=mf=va=0}
   else if ($1 ~ /^scratch_store/) st++; else if ($1 ~ /^scratch_load/) ld++; else if ($1 ~ /^ds_read/) dr++; else if ($1 ~ /^ds_write/) dw++;
   else if ($1 ~ /^v_mfma/) mf++; else if ($1 ~ /^v_/) va++; }
   END {printf("tail: scratch st %d ld %d | ds_read %d ds_write %d mfma %d valu %d\n", st, ld, dr, dw, mf, va)}' $S

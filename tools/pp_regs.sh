#!/bin/bash
# Development: registers / scratch of mel_decoder_pp_kernel<5> and the scratch + instruction histogram between barriers.
D=/tmp/dec_isa_pp; mkdir -p $D; cd $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -Wno-pass-failed -fPIC "$@" -I/root/repo/efficientspeech_amd/csrc -c /root/repo/efficientspeech_amd/csrc/tu_dec_pp.hip -o t.o -save-temps \
   -Rpass-analysis=kernel-resource-usage 2>&1 | tee log.txt | grep -A12 "Function Name: _ZN4esmi21mel_decoder_pp_kernelILi5" | grep -E "VGPRs:|Scratch|Spill"
grep -E " error" -A3 log.txt | head -20
S=$(ls *gfx950.s)
awk '/^_ZN4esmi21mel_decoder_pp_kernelILi5.*: / {on=1} on && /\.Lfunc_end/ {on=0} on {
   if ($1=="s_barrier") {printf("barrier %d: scratch st %d ld %d | ds_read %d ds_write %d mfma %d valu %d\n", nb++, st, ld, dr, dw, mf, va); st=ld=dr=dw=mf=va=0}
   else if ($1 ~ /^scratch_store/) st++; else if ($1 ~ /^scratch_load/) ld++; else if ($1 ~ /^ds_read/) dr++; else if ($1 ~ /^ds_write/) dw++;
   else if ($1 ~ /^v_mfma/) mf++; else if ($1 ~ /^v_/) va++; }
   END {printf("tail: scratch st %d ld %d | ds_read %d ds_write %d mfma %d valu %d\n", st, ld, dr, dw, mf, va)}' $S

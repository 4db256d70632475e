#!/bin/bash
# Development: registers / scratch of the encoder-side chain kernels for a set of -D flags.   tools/chain_regs.sh [-D...]
cd /tmp
for tu in tu_enc_block tu_enc_fuse_va; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -Wno-pass-failed -fPIC "$@" -I/root/repo/efficientspeech_amd/csrc -c /root/repo/efficientspeech_amd/csrc/$tu.hip -o /tmp/chain_regs_$tu.o -Rpass-analysis=kernel-resource-usage 2>&1 \
   | grep -E "Function Name|  VGPRs:|ScratchSize|VGPRs Spill" | sed 's/.*\(Function Name: [^ ]*\).*/\1/; s/.*\(VGPRs: [0-9]*\).*/\1/; s/.*\(ScratchSize[^:]*: [0-9]*\).*/\1/; s/.*\(VGPRs Spill: [0-9]*\).*/\1/' | paste - - - - \
   | grep -E "enc_attn_ffn_split_kernelILi2ELi2ELi1ELi1ELi1ELi2|enc_attn_ffn_kernelILi4ELi1ELi1ELi4ELi3ELi1|enc_fuse_va_kernelILi1ELi3" | sed 's/Function Name: _ZN4esmi[0-9]*//; s/EEvNS.*PE//'
done

// (development, round 4) selects the decoder kernel form: the LayerNorm-folded form (mel_decoder_fold.h) or the row-owner
// LayerNorm form of rounds 1-3 (mel_decoder.h) for A/B measurements: -DESMI_DEC_FOLD=0
#pragma once
#ifndef ESMI_DEC_FOLD
#define ESMI_DEC_FOLD 1
#endif
#if ESMI_DEC_FOLD
#include "mel_decoder_fold.h"
#else
#include "mel_decoder.h"
#endif

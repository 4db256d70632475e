"""Model-size table for the EfficientSpeech acoustic model.

The three published sizes are flag combinations of the reference CLI
(/root/reference/utils/tools.py:344-385, README.md:63,79,189,195); the
published checkpoints use the CLI defaults (decoder kernel 5), not the
`EfficientSpeech.__init__` defaults.
"""
from dataclasses import dataclass

N_SYMBOLS = 152          # len(text.symbols.symbols) in the reference (probed); embedding rows = 153
PAD_ID = 0               # nn.Embedding(padding_idx=0), layers/networks.py:32
HOP_LENGTH = 256         # config/LJSpeech/preprocess.yaml:16
SAMPLING_RATE = 22050    # config/LJSpeech/preprocess.yaml:20

# preprocessed_data/LJSpeech/stats.json ("pitch"[:2], "energy"[:2]) -- bin ranges fed to
# PhonemeEncoder(pitch_stats=, energy_stats=) by model.py:127-139.
LJSPEECH_PITCH_STATS = (-2.9170793047299672, 11.391254536985771)
LJSPEECH_ENERGY_STATS = (-1.431044578552246, 8.184337615966797)


@dataclass(frozen=True)
class ESConfig:
    name: str = "tiny"
    depth: int = 2
    reduction: int = 4
    head: int = 1
    embed_dim: int = 128
    kernel_size: int = 3
    expansion: int = 1
    n_blocks: int = 2
    block_depth: int = 2
    decoder_kernel_size: int = 5
    n_mel_channels: int = 80

    @property
    def dim(self) -> int:                  # model.py:141  dim=embed_dim//reduction
        return self.embed_dim // self.reduction

    @property
    def d4(self) -> int:                   # networks.py:270  dim_x4
        return 4 * self.dim

    @property
    def dx2(self) -> int:                  # networks.py:269  dim_x2
        return min(4 * self.dim, 256)

    @property
    def halo(self) -> int:
        """Receptive-field half width of the mel decoder in frames."""
        return (self.decoder_kernel_size // 2) * self.n_blocks * self.block_depth

    def encoder_kwargs(self):
        return dict(depth=self.depth, reduction=self.reduction, head=self.head,
                    embed_dim=self.embed_dim, kernel_size=self.kernel_size,
                    expansion=self.expansion)

    def decoder_kwargs(self):
        return dict(dim=self.dim, kernel_size=self.decoder_kernel_size,
                    n_mel_channels=self.n_mel_channels, n_blocks=self.n_blocks,
                    block_depth=self.block_depth)


CONFIGS = {
    "tiny": ESConfig(name="tiny"),
    "small": ESConfig(name="small", n_blocks=3, reduction=2),
    "base": ESConfig(name="base", head=2, reduction=1, expansion=2, kernel_size=5,
                     n_blocks=3, block_depth=3),
}

# exact parameter counts probed from the reference
PARAM_COUNTS = {"tiny": 266_417, "small": 952_465, "base": 3_953_489}

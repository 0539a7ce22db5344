"""Architecture / run configuration of the DiffuEraser hot path (reference call site
diffuerase.py:39-45 names the SD-1.5 / sd-vae-ft-mse / diffuEraser checkpoints whose shapes these are)."""
from dataclasses import dataclass, field
from typing import Tuple


@dataclass(frozen=True)
class UNetConfig:
    block_out: Tuple[int, ...] = (320, 640, 1280, 1280)
    attn_levels: Tuple[bool, ...] = (True, True, True, False)   # CrossAttn x3, plain Down (up path mirrors)
    layers_per_block: int = 2
    heads: int = 8
    cross_dim: int = 768
    text_len: int = 77
    groups: int = 32
    in_ch: int = 4
    out_ch: int = 4
    brush_in_ch: int = 9          # 4 noisy latents + 4 masked-image latents + 1 mask
    motion_max_seq: int = 32
    zero_conv_gain: float = 0.5
    # WHERE the BrushNet down residuals enter the UNet ([UNVERIFIED-3P]: the fork's model code is not in the image; VERDICT r5 missing 3).
    #   "skip"   -- added to the skip copies only, after the down path (SURVEY App. D.3's wording; the build's default since round 1);
    #   "hidden" -- added into the running hidden state after conv_in, after every resnet / attention / motion layer and after every downsampler,
    #               so the skip taken there already carries it (how the public BrushNet blocks do it: `hidden_states = hidden_states +
    #               down_block_add_samples.pop(0)` before `output_states += (hidden_states,)`; the conv_in skip is taken BEFORE its residual).
    # One switch, honoured by oracle/model_ref.py::_backbone and unet._Backbone.run_down alike; parity is asserted in both modes
    # (tests/test_model_gpu.py::test_brushnet_residual_site_modes).  The mid and up residuals enter the hidden state in both.
    brushnet_add: str = "skip"

    @property
    def temb_dim(self):
        return 4 * self.block_out[0]


@dataclass(frozen=True)
class VAEConfig:
    block_out: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    groups: int = 32
    latent_ch: int = 4
    scaling: float = 0.18215


# small-but-structurally-complete configs for fast parity tests (same block types, fewer/narrower levels)
TINY_UNET = UNetConfig(block_out=(64, 128), attn_levels=(True, False), layers_per_block=1, heads=2, cross_dim=64,
                       text_len=7, groups=8)
SMALL_UNET = UNetConfig(block_out=(320, 640), attn_levels=(True, False), layers_per_block=1, heads=8, cross_dim=768)
TINY_VAE = VAEConfig(block_out=(32, 64), layers_per_block=1, groups=8)
SMALL_VAE = VAEConfig(block_out=(128, 512), layers_per_block=1)     # mid attention at the real head dim 512


@dataclass(frozen=True)
class RunConfig:
    steps: int = 50
    chunk: int = 32
    overlap: int = 8
    seed: int = 42
    weight_seed: int = 0
    # Defaults = the precision plan that meets the north-star tolerance (per-pixel max-abs <= 1e-3 vs the fp32 oracle at 50 steps,
    # profiles/r2_parity_gpu.txt): fp16 MFMA operands (the reference itself computes in fp16 on GPU) + split-precision VAE decoder.
    # dtype="bf16", precise_decoder=False is ~5 % faster at 1e-2.
    dtype: str = "fp16"           # MFMA operand type: "bf16" | "fp16"
    # time-embedding linears + the UNet's conv_in / conv_out in split precision too (round 4: ~0.1 % of a step, -7 % rms error: tools/parity_rank.py)
    precise_io: bool = True
    precise_decoder: bool = True  # VAE decoder GEMMs as 3 split-precision passes (hi/lo operands): +3 % time at 50 steps, halves the pixel error
    # temporal scheme: "chunks" = independent `chunk`-frame clips + overlap cross-fade (build-defined, shards over GPUs);
    # "reference" = the third-party pipeline's own scheme (22-frame windows shifted on odd steps, value/count averaging, key-frame
    # pre-inference) -- single GPU only
    windowing: str = "chunks"
    # chunks of one rank that are in flight at once, each on its own HIP stream (pipeline.forward_device): the memory-bound kernels of one
    # chunk (GroupNorm, LayerNorm, the fp32-trunk GEMMs) run beside the MFMA-bound kernels of the other and kernel tails overlap.  Chunks are
    # independent until blend time, so the result is bit-identical for every value (tests/test_model_gpu.py); the live set is ~10 GB per chunk
    concurrent_chunks: int = 2
    # lane k starts k * lane_stagger_s seconds after lane 0 (host-side delay before its first launch): identical chunks otherwise run in lockstep,
    # MFMA-bound kernels beside MFMA-bound kernels; an offset puts one chunk's memory-bound phases beside the other's matrix-bound ones
    lane_stagger_s: float = 0.0
    unet: UNetConfig = field(default_factory=UNetConfig)
    vae: VAEConfig = field(default_factory=VAEConfig)

"""Block layout of the reference UNet, restated.

Follows `UNetModel.__init__` (ldm/modules/diffusionmodules/openaimodel.py:443-692)
for the configuration family SD v1 uses: `use_spatial_transformer=True`,
`legacy=False`, `num_head_channels=-1`, `conv_resample=True`,
`resblock_updown=False`, `use_scale_shift_norm=False`, `dims=2`.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    model_channels: int = 320
    attention_resolutions: Tuple[int, ...] = (4, 2, 1)
    num_res_blocks: int = 2
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_heads: int = 8
    transformer_depth: int = 1
    context_dim: int = 768

    def ref_kwargs(self):
        """kwargs for the reference `UNetModel` (configs/stable-diffusion/v1-inference.yaml:29-44)."""
        return dict(image_size=32, in_channels=self.in_channels, out_channels=self.out_channels,
                    model_channels=self.model_channels,
                    attention_resolutions=list(self.attention_resolutions),
                    num_res_blocks=self.num_res_blocks, channel_mult=list(self.channel_mult),
                    num_heads=self.num_heads, use_spatial_transformer=True,
                    transformer_depth=self.transformer_depth, context_dim=self.context_dim,
                    use_checkpoint=False, legacy=False)


SD_V1 = UNetConfig()
# small config used by fast parity tests: same topology (4 levels, attention at
# three of them, 2 res blocks), channel counts that keep every kernel constraint
# of the HIP path (C % 64 == 0, d_head % 8 == 0).
TINY = UNetConfig(model_channels=64, channel_mult=(1, 2, 4, 4), num_heads=2, context_dim=128)
# one more: d_head 40 / 80 like SD at two levels only
SMALL40 = UNetConfig(model_channels=320, channel_mult=(1, 2), attention_resolutions=(2, 1),
                     num_res_blocks=1, num_heads=8, context_dim=256)


@dataclass
class Layer:
    kind: str            # 'conv_in' | 'res' | 'attn' | 'down' | 'up'
    prefix: str          # state_dict prefix, e.g. 'input_blocks.1.0'
    cin: int
    cout: int
    heads: int = 0
    d_head: int = 0


@dataclass
class Plan:
    cfg: UNetConfig
    input_blocks: List[List[Layer]] = field(default_factory=list)
    middle_block: List[Layer] = field(default_factory=list)
    output_blocks: List[List[Layer]] = field(default_factory=list)

    def all_layers(self):
        for blk in self.input_blocks:
            yield from blk
        yield from self.middle_block
        for blk in self.output_blocks:
            yield from blk


def build_plan(cfg: UNetConfig) -> Plan:
    p = Plan(cfg)
    mc = cfg.model_channels
    p.input_blocks.append([Layer('conv_in', 'input_blocks.0.0', cfg.in_channels, mc)])
    chans = [mc]
    ch, ds = mc, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            n = len(p.input_blocks)
            blk = [Layer('res', f'input_blocks.{n}.0', ch, mult * mc)]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                blk.append(Layer('attn', f'input_blocks.{n}.1', ch, ch, cfg.num_heads, ch // cfg.num_heads))
            p.input_blocks.append(blk)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            n = len(p.input_blocks)
            p.input_blocks.append([Layer('down', f'input_blocks.{n}.0', ch, ch)])
            chans.append(ch)
            ds *= 2
    p.middle_block = [
        Layer('res', 'middle_block.0', ch, ch),
        Layer('attn', 'middle_block.1', ch, ch, cfg.num_heads, ch // cfg.num_heads),
        Layer('res', 'middle_block.2', ch, ch),
    ]
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            n = len(p.output_blocks)
            blk = [Layer('res', f'output_blocks.{n}.0', ch + ich, mc * mult)]
            ch = mc * mult
            if ds in cfg.attention_resolutions:
                blk.append(Layer('attn', f'output_blocks.{n}.{len(blk)}', ch, ch, cfg.num_heads, ch // cfg.num_heads))
            if level and i == cfg.num_res_blocks:
                blk.append(Layer('up', f'output_blocks.{n}.{len(blk)}', ch, ch))
                ds //= 2
            p.output_blocks.append(blk)
    return p

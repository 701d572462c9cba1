"""The HOPE actor / critic networks on stock PyTorch-ROCm (BASELINE configs 4 and 5).

Restates the SHAPE of the reference's networks so that the released `HOPE_*.pt` state_dicts load by name and a
random-init network has the same parameter count and initial distribution:

  HopeNet        <-> MultiObsEmbedding   src/model/network.py:34-187  (per-modality embedding MLPs, a 1-layer pre-norm
                                          transformer over the modality tokens, a 2-layer head)
  _TokenMixer    <-> AttentionNetwork    src/model/attention.py:76-94 (Transformer :59-74, Attention :15-42,
                                          FeedForward :44-57, PreNorm :7-13)
  _ImgEncoder    <-> ImgEncoder          src/model/network.py:286-305 (ConvBlock :189-229)
  SacCritic      <-> SACCriticAdapter    src/model/agent/sac_agent.py:15-30 (state + action token, scalar Q)
  ACTOR_CONFIGS / CRITIC_CONFIGS         src/configs.py:131-177

Module attribute names are chosen so that `state_dict()` keys equal the reference's (`embed_lidar.0.weight`,
`net.encoder.layers.0.0.fn.to_qkv.weight`, `net.output.2.bias`, ...).  The forward pass is written for large batches
(N = 10^4..10^5 scenes, 4-5 tokens each): one fused scaled-dot-product call per layer, no python per-sample work.
The policy is plain PyTorch by design (north_star: "the transformer + SAC/PPO policy runs on stock PyTorch-ROCm").
"""
import math
import os

# the image encoder's two small convolutions: take MIOpen's immediate-mode kernels instead of an exhaustive search at
# first use (minutes on a fresh box for every new batch size)
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

LIDAR_NUM, TARGET_DIM, N_DISCRETE_ACTION = 120, 5, 42

ATTENTION_CONFIG = {'depth': 1, 'heads': 8, 'dim_head': 32, 'mlp_dim': 128, 'hidden_dim': 128}   # configs.py:122-128


def net_configs(output_size, use_img=True, use_action_mask=True, use_tanh_output=None, use_attention=True):
    """ACTOR_CONFIGS (output_size 2, tanh output) / CRITIC_CONFIGS (output_size 1, linear output), configs.py:131-177."""
    return {
        'n_modal': 2 + int(use_img) + int(use_action_mask),
        'lidar_shape': LIDAR_NUM, 'target_shape': TARGET_DIM,
        'action_mask_shape': N_DISCRETE_ACTION if use_action_mask else None,
        'img_shape': (3, 64, 64) if use_img else None,
        'output_size': output_size, 'embed_size': 128, 'hidden_size': 256, 'n_hidden_layers': 3, 'n_embed_layers': 2,
        'img_conv_layers': [4, 8], 'img_linear_layers': [256], 'k_img_conv': 3, 'orthogonal_init': True,
        'use_tanh_output': (output_size == 2) if use_tanh_output is None else use_tanh_output,
        'use_tanh_activate': True,
        'attention_configs': dict(ATTENTION_CONFIG) if use_attention else None,
    }


def actor_configs(**kw):
    return net_configs(2, **kw)


def critic_configs(**kw):
    return net_configs(1, **kw)


def _act(use_tanh):
    return nn.Tanh() if use_tanh else nn.LeakyReLU()


def _embed_mlp(n_in, width, n_layers, use_tanh):
    """Linear(n_in, width) followed by (n_layers - 1) x [act, Linear(width, width)] (network.py:69-73)."""
    mods = [nn.Linear(n_in, width)]
    for _ in range(n_layers - 1):
        mods += [_act(use_tanh), nn.Linear(width, width)]
    return nn.Sequential(*mods)


class _SelfAttention(nn.Module):
    """multi-head self-attention over the modality tokens, bias-free fused qkv (attention.py:15-42)."""

    def __init__(self, dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        self.to_qkv = nn.Linear(dim, 3 * inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim)) if not (heads == 1 and dim_head == dim) else nn.Identity()

    def forward(self, x):                                   # x [B, T, dim]
        b, t, _ = x.shape
        qkv = self.to_qkv(x).view(b, t, 3, self.heads, self.dim_head).permute(2, 0, 3, 1, 4)    # [3, B, H, T, d]
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])                               # scale = d ** -0.5
        return self.to_out(o.transpose(1, 2).reshape(b, t, self.heads * self.dim_head))


class _TokenFF(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        # indices 0 and 3 carry the parameters, as in the reference's Sequential(Linear, Tanh, Dropout, Linear, Dropout)
        self.net = nn.Sequential(nn.Linear(dim, hidden), nn.Tanh(), nn.Identity(), nn.Linear(hidden, dim))

    def forward(self, x):
        return self.net(x)


class _Pre(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x):
        return self.fn(self.norm(x))


class _Encoder(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim):
        super().__init__()
        self.layers = nn.ModuleList([nn.ModuleList([_Pre(dim, _SelfAttention(dim, heads, dim_head)),
                                                    _Pre(dim, _TokenFF(dim, mlp_dim))]) for _ in range(depth)])

    def forward(self, x):
        for attn, ff in self.layers:
            x = x + attn(x)
            x = x + ff(x)
        return x


class _TokenMixer(nn.Module):
    """AttentionNetwork (attention.py:76-94): transformer over [B, n_tokens, dim], tokens concatenated, 2-layer head.
    `view_embed` exists (it is in the checkpoints) but the reference never adds it (:90)."""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, n_tokens, hidden_dim, output_dim):
        super().__init__()
        self.encoder = _Encoder(dim, depth, heads, dim_head, mlp_dim)
        self.output = nn.Sequential(nn.Linear(n_tokens * dim, hidden_dim), nn.Tanh(), nn.Linear(hidden_dim, output_dim))
        self.view_embed = nn.Parameter(torch.zeros(1, n_tokens, dim))

    def forward(self, x):
        return self.output(self.encoder(x).flatten(1))


class _ConvBlock(nn.Module):
    """conv KxK + act + maxpool 2, plus a 1x1-conv / avgpool-2 shortcut (network.py:189-229, Batch_norm=False)."""

    def __init__(self, cin, cout, k, use_tanh):
        super().__init__()
        self.layer = nn.Sequential(nn.Conv2d(cin, cout, kernel_size=k, padding=k // 2), _act(use_tanh), nn.MaxPool2d(2))
        self.shortcut = nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1), nn.AvgPool2d(2))

    def forward(self, x):
        return self.layer(x) + self.shortcut(x)


class _ImgEncoder(nn.Module):
    def __init__(self, shape, k, embed, convs, fcs, use_tanh=True):
        super().__init__()
        cin, w, h = shape
        mods, c = [], cin
        for cout in convs:
            mods.append(_ConvBlock(c, cout, k, use_tanh))
            c = cout
        mods.append(nn.Flatten())
        n = (w * h * convs[-1]) // (4 ** len(convs))
        for f in fcs:
            mods += [nn.Linear(n, f), _act(use_tanh)]
            n = f
        self.net = nn.Sequential(*mods)
        self.output_mean = nn.Linear(n, embed)
        self.output_std = nn.Linear(n, embed)           # in the checkpoints; unused by the policy (network.py:180)
        self.amp = False                                # set_img_amp(): bf16 autocast + channels_last for the conv stack

    def forward(self, x):
        if self.amp and x.is_cuda:                      # stock PyTorch levers only (inference-side report, not the default)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                y = self.net(x.contiguous(memory_format=torch.channels_last))
            return self.output_mean(y.float())
        return self.output_mean(self.net(x))


def set_img_amp(module, on=True):
    """bf16 autocast + channels_last for every image encoder under `module` (the reference runs them in fp32; this changes the
    numerics of the policy, not of the env: off by default, `bench.py --policy-amp` reports it)"""
    for m in module.modules():
        if isinstance(m, _ImgEncoder):
            m.amp = bool(on)
            if on:
                m.net.to(memory_format=torch.channels_last)


class HopeNet(nn.Module):
    """obs dict {'lidar' [B,120], 'target' [B,5], 'action_mask' [B,42], 'img' [B,3,64,64] (uint8 or float in [0,1]),
    optional 'action' [B,2]} -> [B, output_size]."""

    def __init__(self, cfg, img_use_tanh=True):
        super().__init__()
        e, tanh = cfg['embed_size'], bool(cfg['use_tanh_activate'])
        self.cfg = dict(cfg)
        self.use_img = cfg['img_shape'] is not None
        self.use_action_mask = cfg['action_mask_shape'] is not None
        self.input_action = cfg.get('input_action_dim', 0) > 0
        att = cfg['attention_configs']
        self.use_attention = att is not None
        if self.use_attention:
            self.net = _TokenMixer(e, att['depth'], att['heads'], att['dim_head'], att['mlp_dim'], cfg['n_modal'],
                                   att['hidden_dim'], cfg['output_size'])
        else:                                                       # network.py:45-53 (note: no activation after the last hidden)
            nh, hs = cfg['n_hidden_layers'], cfg['hidden_size']
            if nh == 1:
                mods = [nn.Linear(cfg['n_modal'] * e, cfg['output_size'])]
            else:
                mods = [nn.Linear(cfg['n_modal'] * e, hs)]
                for _ in range(nh - 2):
                    mods += [_act(tanh), nn.Linear(hs, hs)]
                mods.append(nn.Linear(hs, cfg['output_size']))
            self.net = nn.Sequential(*mods)
        self.tanh_out = bool(cfg['use_tanh_output'])
        nl = cfg['n_embed_layers']
        self.embed_lidar = _embed_mlp(cfg['lidar_shape'], e, nl, tanh)
        self.embed_tgt = _embed_mlp(cfg['target_shape'], e, nl, tanh)
        if self.use_action_mask:
            self.embed_am = _embed_mlp(cfg['action_mask_shape'], e, nl, tanh)
        if self.use_img:
            self.embed_img = _ImgEncoder(cfg['img_shape'], cfg['k_img_conv'], e, cfg['img_conv_layers'],
                                         cfg['img_linear_layers'], use_tanh=img_use_tanh)
            self.re_embed_img = nn.Sequential(_act(tanh), nn.Linear(e, e))
        if self.input_action:
            self.embed_action = _embed_mlp(cfg['input_action_dim'], e, nl, tanh)
        self.reset_parameters()

    def reset_parameters(self):
        """MultiObsEmbedding.orthogonal_init (network.py:106-160): every >=2-D weight of the trunk and of the
        embedding MLPs orthogonal with gain 1 (the reference's `i` never advances, so the 0.01 output gain is never
        used), their biases 0; the image encoder keeps PyTorch's default init."""
        groups = [self.net, self.embed_lidar, self.embed_tgt]
        groups += [self.embed_am] if self.use_action_mask else []
        groups += [self.re_embed_img] if self.use_img else []
        groups += [self.embed_action] if self.input_action else []
        for g in groups:
            for name, p in g.named_parameters():
                if name.endswith('weight') and p.dim() > 1:
                    nn.init.orthogonal_(p, gain=1.0)
                elif name.endswith('bias'):
                    nn.init.zeros_(p)

    def tokens(self, x):
        f = [self.embed_lidar(x['lidar']), self.embed_tgt(x['target'])]
        if self.use_action_mask:
            f.append(self.embed_am(x['action_mask']))
        if self.use_img:
            img = x['img']
            if img.dtype == torch.uint8:                 # the C ABI hands obs['img'] * 255 as uint8
                img = img.to(f[0].dtype) * (1.0 / 255.0)
            f.append(self.re_embed_img(self.embed_img(img)))
        if self.input_action:
            f.append(self.embed_action(x['action']))
        return f

    def forward(self, x):
        f = self.tokens(x)
        out = self.net(torch.stack(f, dim=1) if self.use_attention else torch.cat(f, dim=1))
        return torch.tanh(out) if self.tanh_out else out


class SacCritic(nn.Module):
    """Q(s, a): the critic config plus one more token for the action (sac_agent.py:15-30)."""

    def __init__(self, cfg, action_dim=2, img_use_tanh=True):
        super().__init__()
        c = dict(cfg)
        c['input_action_dim'] = action_dim
        c['n_modal'] = cfg['n_modal'] + 1
        self.net = HopeNet(c, img_use_tanh=img_use_tanh)

    def forward(self, state, action):
        x = dict(state)
        x['action'] = action
        return self.net(x)


def count_parameters(module):
    return sum(p.numel() for p in module.parameters())


def gaussian_log_prob(mean, log_std, action):
    """Normal(mean, exp(log_std)).log_prob(action), per action dimension (ppo_agent.py:137-140,317-321)."""
    var2 = torch.exp(2.0 * log_std)
    return -((action - mean) ** 2) / (2.0 * var2) - log_std - 0.5 * math.log(2.0 * math.pi)

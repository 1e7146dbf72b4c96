"""DeiT-base-distilled-384 trunk with per-block outputs.

Mirror of FusionTransformer/models/transformers.py:11-45,90-100.  timm is not a
dependency: the VisionTransformer structure of timm 0.4.9 (PatchEmbed, Block,
Attention, Mlp; LayerNorm eps 1e-6, exact GELU, qkv_bias=True, distilled) is
restated here with the same attribute names, so DeiT checkpoints keyed
`backbone.blocks.0.attn.qkv.weight` etc. load unchanged.

The dense contractions (patch-embed conv, qkv / proj / MLP linears) go through
torch's hipBLASLt GEMMs; softmax(QK^T/8)V runs in libftx's fused attention
kernel when `attn_impl == "ftx"`, otherwise as the three explicit ops timm uses."""
from __future__ import annotations

import os
from functools import partial
from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ["Image2DTransformer", "image_2d_distilled_transformer"]


_ONES = {}


def _ones_row(m, like):
    key = (m, like.device, like.dtype)
    if key not in _ONES:
        _ONES[key] = torch.ones(1, m, device=like.device, dtype=like.dtype)
    return _ONES[key]


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b with the bias gradient from libftx's column-sum kernels (ftx_colsum) instead of autograd's sum_to reduction:
    that reduction kernel's result changes between the first and the later replays of a captured backward on torch 2.10 / ROCm 7
    (tools/probes/graph_block4.py), so it cannot stay in a trunk that replays as a HIP graph.  Round 1 used a (1, M) x (M, N) library
    GEMM for it (14 us per Linear at batch 4, 48 per step); the two column-sum launches take 3-6 us."""

    @staticmethod
    def forward(ctx, x, w, b, bf16=False):
        # b None: the caller adds the bias itself (Block.chain hands it to the fused add + LayerNorm that consumes this output)
        x2 = x.reshape(-1, x.shape[-1])
        ctx.in_shape, ctx.bf16 = x.shape, bool(bf16)
        if bf16:
            # BASELINE configs[4] ("bf16 forward"): operands rounded to bf16, products accumulated in fp32 by the bf16 MFMA
            # path, result and bias add in fp32; the two backward GEMMs run the same way, the bias gradient stays fp32
            x2, w = x2.to(torch.bfloat16), w.to(torch.bfloat16)
            ctx.save_for_backward(x2, w)
            y = (x2 @ w.t()).float()
            return (y.add_(b) if b is not None else y).view(*x.shape[:-1], w.shape[0])
        ctx.save_for_backward(x2, w)
        y = torch.addmm(b, x2, w.t()) if b is not None else x2 @ w.t()
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        db = None
        if ctx.needs_input_grad[2]:
            # column sums by libftx (float64, fixed order); a width that is not a multiple of 4 goes to the skinny GEMM
            if dy2.shape[1] % 4 == 0:
                from .. import functional as spf
                db = spf.colsum(dy2)
            else:
                db = (_ones_row(dy2.shape[0], dy2) @ dy2).view(-1)
        if ctx.bf16:
            dy2 = dy2.to(torch.bfloat16)
        dx = (dy2 @ w).float().view(ctx.in_shape) if ctx.needs_input_grad[0] else None
        dw = (dy2.t() @ x2).float() if ctx.needs_input_grad[1] else None
        return dx, dw, db, None


class _BatchBroadcast(torch.autograd.Function):
    """p (1, ...) -> p expanded over a batch of b; the gradient (the sum over the batch) by ftx_colsum instead of autograd's sum_to
    reduction, which does not replay reliably inside a captured backward (see _LinearFn)."""

    @staticmethod
    def forward(ctx, p, b):
        ctx.pshape = p.shape
        return p.expand(b, *p.shape[1:])

    @staticmethod
    def backward(ctx, g):
        from .. import functional as spf
        b = g.shape[0]
        g2 = g.reshape(b, -1)
        if g2.is_cuda and g2.dtype == torch.float32 and g2.shape[1] % 4 == 0:
            return spf.colsum(g2).view(ctx.pshape), None
        return g2.sum(0).view(ctx.pshape), None


def _over_batch(p, b):
    return _BatchBroadcast.apply(p, b) if p.is_cuda else p.expand(b, *p.shape[1:])


def _linear(x, lin, with_bias=True):
    if lin.bias is None or not x.is_cuda:
        return F.linear(x, lin.weight, lin.bias if with_bias else None)
    return _LinearFn.apply(x, lin.weight, lin.bias if with_bias else None, getattr(lin, "ftx_bf16", False))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x, with_fc2_bias=True):
        return self.drop(_linear(self.drop(self.act(_linear(x, self.fc1))), self.fc2, with_fc2_bias))


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.attn_impl = "ftx"

    def forward(self, x, with_proj_bias=True):
        B, N, C = x.shape
        qkv = _linear(x, self.qkv)
        if self.attn_impl == "ftx":
            from .. import functional as spf
            x = spf.attention(qkv.view(B, N, 3, self.num_heads, C // self.num_heads), self.scale)
        else:
            qkv = qkv.reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
            q, k, v = qkv[0], qkv[1], qkv[2]
            attn = (q @ k.transpose(-2, -1)) * self.scale
            attn = attn.softmax(dim=-1)
            attn = self.attn_drop(attn)
            x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(_linear(x, self.proj, with_proj_bias))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        if not self._fused(x):
            x = x + self.drop_path(self.attn(self.norm1(x)))
            x = x + self.drop_path(self.mlp(self.norm2(x)))
            return x
        return _materialize(*self.chain(x, None, None))

    def _fused(self, x):
        from .. import functional as spf
        return (spf.layer_norm_supported(x) and type(self.norm1) is nn.LayerNorm and type(self.norm2) is nn.LayerNorm
                and self.norm1.elementwise_affine and self.norm2.elementwise_affine and self.norm1.bias is not None and self.norm2.bias is not None
                and self.attn.proj.bias is not None and self.mlp.fc2.bias is not None and self.attn.proj_drop.p == 0.0 and self.mlp.drop.p == 0.0
                and isinstance(self.drop_path, nn.Identity))

    def chain(self, r, p, pb):
        """The block on a residual stream held as (r, p, pb) with value r + (p + pb) (p None: just r; pb: the bias that the Linear
        which produced p did NOT add); returns it in the same form, (s2, m, fc2.bias) with the MLP output m not yet added and
        computed without its bias (`_materialize` turns the triple into one tensor, in the same rounding order).  Each LayerNorm runs fused with the add in front of it (libftx ftx_add_layernorm_*): the
        add of the previous block's MLP output with norm1, the add of the attention output with norm2 -- same arithmetic in the same
        rounding order, a third of the launches (forward: add + LayerNorm -> 1; backward: 3 LayerNorm-gradient kernels + the
        residual-gradient add + the column sums for the proj / fc2 bias gradients -> 2)."""
        from .. import functional as spf
        n1, n2 = self.norm1, self.norm2
        if p is None:
            s, h = r, spf.layer_norm(r, n1.weight, n1.bias, n1.eps)
        else:
            s, h = spf.add_layer_norm(r, p, n1.weight, n1.bias, n1.eps, y_bias=pb)
        s2, h2 = spf.add_layer_norm(s, self.attn(h, with_proj_bias=False), n2.weight, n2.bias, n2.eps, y_bias=self.attn.proj.bias)
        return s2, self.mlp(h2, with_fc2_bias=False), self.mlp.fc2.bias


class _Materialize(torch.autograd.Function):
    """r + (p + pb) with pb broadcast over the rows; the bias gradient by ftx_colsum (float64, fixed order) instead of autograd's sum_to
    reduction, which must not run inside a captured backward (see _LinearFn)."""

    @staticmethod
    def forward(ctx, r, p, pb):
        return r + (p + pb)

    @staticmethod
    def backward(ctx, g):
        gb = None
        if ctx.needs_input_grad[2]:
            g2 = g.reshape(-1, g.shape[-1])
            if g2.is_cuda and g2.dtype == torch.float32 and g2.shape[1] % 4 == 0:
                from .. import functional as spf
                gb = spf.colsum(g2)
            else:
                gb = g2.sum(0)
        return g, g, gb


def _materialize(r, p, pb):
    """r + (p + pb): the residual stream of Block.chain as one tensor (bias first, as a GEMM's bias epilogue rounds it)."""
    if p is None:
        return r
    return _Materialize.apply(r, p, pb)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.Identity()

    def forward(self, x):
        # Conv2d(k=16, s=16) == non-overlapping unfold + one GEMM (hipBLASLt); same arithmetic as
        # timm's self.proj(x).flatten(2).transpose(1, 2) without a convolution-library search
        B, C, H, W = x.shape
        ph, pw = self.patch_size
        gh, gw = H // ph, W // pw
        patches = x.reshape(B, C, gh, ph, gw, pw).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * ph * pw)
        w2 = self.proj.weight.view(self.proj.out_channels, -1)
        if x.is_cuda and self.proj.bias is not None and self.proj.out_channels % 4 == 0:
            return self.norm(_LinearFn.apply(patches, w2, self.proj.bias, False))    # bias gradient by ftx_colsum, not autograd's sum_to (see _LinearFn)
        return self.norm(F.linear(patches, w2, self.proj.bias))


class Image2DTransformer(nn.Module):
    def __init__(self, remove_tokens_outputs=False, img_size=384, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, qkv_bias=True, distilled=True, last_block=None, **_unused):
        super().__init__()
        self.remove_tokens_outputs = remove_tokens_outputs
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 2 if distilled else 1
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.dist_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if distilled else None
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + self.num_tokens, embed_dim))
        self.pos_drop = nn.Dropout(p=0.0)
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)   # kept for checkpoint compatibility; forward_blocks never applies it
        self.head = nn.Identity()           # reset_classifier(0, '') in image_models_billinear.py:44,57
        self.head_dist = nn.Identity()
        # Blocks after `last_block` never influence an output the model returns
        # (SURVEY Appendix A.12); they are skipped when it is set.
        self.last_block = last_block
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        if self.dist_token is not None:
            nn.init.trunc_normal_(self.dist_token, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.zeros_(m.bias)
                nn.init.ones_(m.weight)

    def set_bf16(self, on: bool = True):
        """bf16 operands for the qkv / proj / MLP GEMMs of every block (fp32 accumulate, fp32 everywhere else: LayerNorm,
        softmax, residual stream, attention kernel).  No counterpart in the reference, which is fp32 end to end; the parity
        bar for this mode is stated in tests/test_model_gpu.py."""
        for blk in self.blocks:
            for lin in (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2):
                lin.ftx_bf16 = bool(on)

    def set_attention_impl(self, impl: str):
        for blk in self.blocks:
            blk.attn.attn_impl = impl

    def _embed(self, x):
        x = self.patch_embed(x)
        b = x.shape[0]
        cls_token = _over_batch(self.cls_token, b)
        if self.dist_token is None:
            x = torch.cat((cls_token, x), dim=1)
        else:
            x = torch.cat((cls_token, _over_batch(self.dist_token, b), x), dim=1)
        return self.pos_drop(x + _over_batch(self.pos_embed, b))

    def forward_blocks(self, x: torch.Tensor, on_block=None) -> Dict[str, torch.Tensor]:
        """reference models/transformers.py:16-45: every block's output, cls/dist tokens stripped.
        `on_block(i, tokens)` is called as soon as block i's output exists (used to lift the tapped
        block's features while the remaining blocks are still being issued).

        With `graph_taps` set (training on the GPU) the trunk runs as one HIP graph per tapped segment --
        forward and backward: the shapes are static, and ~500 kernel launches per step become 2 x 2 graph
        launches.  Only the tapped blocks' outputs are returned then (the model reads no others)."""
        graphed = self._graphed_segments(x)
        if graphed is not None:
            outputs = dict()
            for last, seg, is_tap in graphed:
                x = seg(x)
                if is_tap:
                    outputs[str(last)] = x[:, self.num_tokens:, :] if self.remove_tokens_outputs else x
                    if on_block is not None:
                        on_block(last, outputs[str(last)])
            return outputs
        infer = self._inference_graph(x)
        if infer is not None:
            # validate()'s forward (data/utils/validate.py:59): the whole live trunk as ONE forward-only HIP graph; the tapped outputs
            # are copies, so a later replay does not change tensors the caller still holds
            outputs = dict()
            for i, t in infer(x):
                outputs[str(i)] = t[:, self.num_tokens:, :] if self.remove_tokens_outputs else t
                if on_block is not None:
                    on_block(i, outputs[str(i)])
            return outputs
        if x.is_cuda and self.training and torch.is_grad_enabled():
            # This pass and its backward run eagerly.  A capture attempted AFTER an eager backward through these parameters
            # aborts the process inside hipStreamEndCapture on torch 2.10 / ROCm 7 (tools/probes/graph_recapture.py) -- an abort,
            # not an exception -- so from here on no NEW capture is attempted (already captured shapes keep replaying).
            self._capture_off("an eager training pass ran through the trunk before the capture")
        outputs = dict()

        def emit(i, t):
            outputs[str(i)] = t[:, self.num_tokens:, :] if self.remove_tokens_outputs else t
            if on_block is not None:
                on_block(i, outputs[str(i)])

        self._forward_eager(x, emit)
        return outputs

    def _forward_eager(self, x, emit, want=None):
        """The live blocks, eagerly; emit(i, tokens) for every block i (want=None: what the reference's forward_blocks returns) or only
        for the blocks in `want`.

        Same residual-stream form as the captured segments (_TrunkSegment): inside a segment the stream stays (r, p, pb) from block to
        block, at a segment end it becomes one tensor -- so the eager trunk and the graphed trunk run the SAME kernels in the same order
        and agree to the last bit in every gradient.  A block's output in between is materialised for the caller only."""
        x = self._embed(x)
        live = [b for i, b in enumerate(self.blocks) if self.last_block is None or i <= self.last_block]
        ends = set(self._segment_ends()) if self.graph_taps else None      # no taps known: every block ends a segment
        chained = x.is_cuda and ends is not None and all(b._fused(x) for b in live)
        r, p, pb = x, None, None
        for i, block in enumerate(live):
            need = want is None or i in want
            if chained:
                r, p, pb = block.chain(r, p, pb)
                if i in ends:
                    x = _materialize(r, p, pb)
                    r, p, pb = x, None, None
                elif need:
                    x = _materialize(r, p, pb)
            else:
                x = block(x)
            if need:
                emit(i, x)

    # ---- HIP-graph execution of the trunk ------------------------------------------------------------------
    graph_taps = None   # sorted block indices whose outputs the caller uses; None: eager execution
    graph_segment_blocks = int(os.environ.get("FTX_GRAPH_SEGMENT_BLOCKS", "3"))   # blocks per captured segment (a tap always ends one)

    def _graph_key(self, x):
        """Everything a captured graph bakes in: input shape / dtype / requires_grad, the tap set and segment length, the
        per-block execution flags and the parameters' requires_grad pattern.  Changing any of them selects another graph."""
        flags = tuple((blk.attn.attn_impl, bool(getattr(blk.attn.qkv, "ftx_bf16", False)), bool(getattr(blk.mlp.fc1, "ftx_bf16", False)))
                      for blk in self.blocks)
        grads = tuple(p.requires_grad for p in self.parameters())
        return (tuple(x.shape), x.dtype, bool(x.requires_grad), tuple(self.graph_taps), self.last_block, self.graph_segment_blocks,
                flags, grads, torch.cuda.current_device())

    def _graphed_segments(self, x):
        if not self.graph_taps or not self.use_graphs or not x.is_cuda or not self.training or not torch.is_grad_enabled():
            return None
        cache = self.__dict__.setdefault("_graph_cache", {})
        key = self._graph_key(x)
        if key not in cache:
            if self.__dict__.get("_graph_capture_off", False):
                # an eager training pass or a refused capture came before: never capture again (see forward_blocks); say so ONCE
                if not self.__dict__.get("_graph_off_reported", False):
                    import sys
                    self.__dict__["_graph_off_reported"] = True
                    print("[fusiontransformer_amd] ViT trunk runs eagerly for input %s: HIP-graph capture is off (%s)"
                          % (tuple(x.shape), self.__dict__.get("_graph_off_reason", "unknown")), file=sys.stderr, flush=True)
                return None
            try:
                cache[key] = self._capture_segments(x)
            except Exception as err:   # capture refused (another thread touched the device, unsupported op, ...): run eagerly
                import sys
                print("[fusiontransformer_amd] HIP-graph capture of the ViT trunk failed (%s: %s); running it eagerly from now on" % (type(err).__name__, err),
                      file=sys.stderr, flush=True)
                cache[key] = None
                self._capture_off("capture refused: %s: %s" % (type(err).__name__, err))
        return cache[key]

    # ---- forward-only graph for evaluation -----------------------------------------------------------------------------------
    eval_graphs = os.environ.get("FTX_VIT_EVAL_GRAPHS", "1") != "0"

    def _inference_graph(self, x):
        """Replay function of the forward-only HIP graph for this input shape, or None (not on the GPU, gradients enabled, training
        mode, taps unknown, or the capture was refused).  No autograd node is created under no_grad, so the restriction of the
        training capture (never after an eager backward, see forward_blocks) does not apply."""
        if (not self.eval_graphs or not self.graph_taps or not self.use_graphs or not x.is_cuda or self.training or torch.is_grad_enabled()
                or torch.cuda.is_current_stream_capturing()):
            return None
        cache = self.__dict__.setdefault("_infer_cache", {})
        flags = tuple((blk.attn.attn_impl, bool(getattr(blk.attn.qkv, "ftx_bf16", False))) for blk in self.blocks)
        key = (tuple(x.shape), x.dtype, tuple(self.graph_taps), self.last_block, flags, torch.cuda.current_device())
        if key not in cache:
            try:
                cache[key] = self._capture_inference(x)
            except Exception as err:     # refused: run eagerly from now on for this shape
                import sys
                print("[fusiontransformer_amd] forward-only HIP-graph capture of the ViT trunk failed (%s: %s); evaluation runs it eagerly" % (type(err).__name__, err),
                      file=sys.stderr, flush=True)
                cache[key] = None
        return cache[key]

    def _capture_inference(self, x):
        taps = sorted(int(t) for t in self.graph_taps)
        static_in = x.detach().clone()

        def run(inp):
            got = {}
            self._forward_eager(inp, lambda i, t: got.__setitem__(i, t), want=set(taps))
            return [got[t] for t in taps]

        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                run(static_in)                                  # warm-up: lazy initialisation, TunableOp selections, scratch growth
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            static_out = run(static_in)

        def replay(inp):
            static_in.copy_(inp)
            graph.replay()
            return [(t, o.clone()) for t, o in zip(taps, static_out)]

        return replay

    def _capture_off(self, reason):
        if not self.__dict__.get("_graph_capture_off", False):
            self.__dict__["_graph_capture_off"] = True
            self.__dict__["_graph_off_reason"] = str(reason)

    def graph_state(self):
        """"on" (every shape seen so far replays as HIP graphs), "off:<reason>" (capture was switched off: later shapes run eagerly,
        at a cost in speed only), "eager" (graph_taps not set), "idle" (nothing captured yet)."""
        if not self.graph_taps or not self.use_graphs:
            return "eager"
        if self.__dict__.get("_graph_capture_off", False):
            return "off:" + self.__dict__.get("_graph_off_reason", "unknown")
        cache = self.__dict__.get("_graph_cache") or {}
        if not cache:
            return "idle"
        return "on" if all(v is not None for v in cache.values()) else "off:a capture fell back to eager execution"

    use_graphs = True    # False: same segment structure, executed eagerly (the twin of bench.py's selfcheck and of the bit-identity tests)

    def _segment_ends(self):
        """Last block of every trunk segment: every tap, and at most `graph_segment_blocks` blocks per segment -- the gradients of a
        segment become available together when its backward graph has run, so shorter segments keep the data-parallel bucket
        all-reduces (dist.GradReducer) overlapped with the rest of the backward."""
        ends, first = [], 0
        for t in sorted(int(t) for t in self.graph_taps):
            while t - first + 1 > self.graph_segment_blocks:
                ends.append(first + self.graph_segment_blocks - 1)
                first = ends[-1] + 1
            ends.append(t)
            first = t + 1
        return ends

    def _capture_segments(self, x):
        taps = sorted(int(t) for t in self.graph_taps)
        ends = self._segment_ends()
        segments, first = [], 0
        for e in ends:
            segments.append(_TrunkSegment(self, first, e, embed=(first == 0)))
            first = e + 1
        # sample inputs with the live requires_grad pattern (the resampled image carries the gradient of sample_down)
        samples = [(x.detach().clone().requires_grad_(x.requires_grad),)]
        with torch.no_grad():
            h = segments[0](samples[0][0])
            for seg in segments[1:]:
                samples.append((h.detach().clone().requires_grad_(True),))
                h = seg(h)
        graphed = torch.cuda.make_graphed_callables(tuple(segments), tuple(samples), num_warmup_iters=3)
        taps_set = set(taps)
        return [(e, g, e in taps_set) for e, g in zip(ends, graphed)]


class _TrunkSegment(nn.Module):
    """Blocks first..last of a trunk (plus the embedding for the first segment) as one capturable callable.  It shares
    the trunk's modules and parameters; it is never registered in the model, so state_dict keys do not change."""

    def __init__(self, trunk, first, last, embed):
        super().__init__()
        self.embed = embed
        if embed:
            object.__setattr__(self, "_trunk", trunk)
            self.patch_embed, self.cls_token, self.dist_token, self.pos_embed = trunk.patch_embed, trunk.cls_token, trunk.dist_token, trunk.pos_embed
        self.blocks = nn.ModuleList([trunk.blocks[i] for i in range(first, last + 1)])

    def forward(self, x):
        if self.embed:
            x = self._trunk._embed(x)
        if not all(block._fused(x) for block in self.blocks):
            for block in self.blocks:
                x = block(x)
            return x
        r, p, pb = x, None, None    # residual stream as (r, p, pb): the add of a block's MLP output (and of fc2's bias) runs inside the next block's norm1 kernel
        for block in self.blocks:
            r, p, pb = block.chain(r, p, pb)
        return _materialize(r, p, pb)


def image_2d_distilled_transformer(pretrained=False, **kwargs):
    """reference models/transformers.py:90-100 (DeiT-base distilled, patch 16, 384x384).

    `pretrained=True` would fetch weights by name; there is no network here, so it is refused."""
    if pretrained:
        raise RuntimeError("pretrained DeiT weights cannot be fetched offline; pass IMAGE_PRETRAINED_PATH or use random init")
    model_kwargs = dict(patch_size=16, embed_dim=768, depth=12, num_heads=12, img_size=384, distilled=True)
    model_kwargs.update(kwargs)
    return Image2DTransformer(**model_kwargs)

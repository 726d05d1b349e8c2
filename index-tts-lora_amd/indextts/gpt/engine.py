"""GPT-2 acoustic-token decoder engine on the HIP kernels (prefill, cached decode loop, teacher-forced latent pass).

Replaces, for the inference path, HF `GPT2Model` + `GenerationMixin.generate` as wired by the reference in
indextts/gpt/model.py:125-205 (GPT2InferenceModel.forward), :263-286 (build_hf_gpt_transformer) and :459-474
(get_logits).  Host code only does tensor plumbing (torch device memory, streams, CUDA-graph capture/replay); every
FLOP of the transformer runs in libindextts_hip.so.

Data layout in HBM
  * weights: packed MFMA B-fragment blocks (include/indextts_hip.h), dtype T (fp32 parity mode / bf16 speed mode);
    LayerNorm parameters, biases and embedding tables stay fp32.
  * residual stream h: fp32 [rows][D].  T-typed scratch: xn, qkv, attn, ffn.
  * KV cache: T [L][Bmax][H][Smax][64], K and V separate; all rows of a (left-padded) batch share one write position.
  * decode-loop state (int32[8], device): step k, cache position, finished count, arrival counter.
"""
from __future__ import annotations

import os

import torch

from .. import _native as nat


KV_TAB = 64   # ITTS_KV_TAB: ring entries of a row's block table


class PagedKV:
    """Paged KV cache of one engine (include/indextts_hip.h, "Paged KV cache"): a block pool per layer, K and V, T
    [L][blocks][H][bs][64]; a block table int32 [rows][64] on the device with a host mirror; a free list.  Block 0 is the
    scratch block: every table entry that maps no live position points at it (a row that has stopped keeps appending its
    stop-token keys there; masked / clamped reads stay inside the pool).  Positions are the decode loop's global positions
    (left-padded rows, one shared write position); the table is a ring over position / bs, so the position counter may grow
    without bound while every row's live window stays under (64 - 2) * bs positions."""

    def __init__(self, L, H, rows, blocks, bs, dtype, device):
        assert bs in (16, 32, 64)
        self.bs, self.shift, self.rows, self.blocks = bs, bs.bit_length() - 1, rows, blocks
        self.kc = torch.zeros(L, blocks, H, bs, 64, dtype=dtype, device=device)
        self.vc = torch.zeros(L, blocks, H, bs, 64, dtype=dtype, device=device)
        self.tab = torch.zeros(rows, KV_TAB, dtype=torch.int32, device=device)
        self.reset()

    def reset(self):
        import numpy as np
        self.tab_h = np.zeros((self.rows, KV_TAB), dtype=np.int32)
        self.free = list(range(self.blocks - 1, 0, -1))     # block 0 = scratch
        self.span = [None] * self.rows                       # per row: [first, last + 1) block INDEX (position >> shift) mapped
        self.dirty = True

    @property
    def window(self):
        """Most positions a row may keep live."""
        return (KV_TAB - 2) * self.bs

    def cover(self, row, lo, hi):
        """Map blocks for positions [lo, hi) of `row` (extends the row's span at its upper end; a new row starts a span)."""
        b0, b1 = lo >> self.shift, ((hi - 1) >> self.shift) + 1
        sp = self.span[row]
        if sp is None:
            sp = self.span[row] = [b0, b0]
        if b1 - sp[0] > KV_TAB - 1:
            raise ValueError(f"paged KV: row {row} would keep {(b1 - sp[0]) * self.bs} positions live (limit {self.window})")
        need = b1 - sp[1]
        if need > 0:
            if need > len(self.free):
                raise RuntimeError("paged KV: the block pool is exhausted")
            import numpy as np
            ids = self.free[-need:][::-1]                   # the blocks pop() would hand out, in that order
            del self.free[-need:]
            self.tab_h[row, (np.arange(sp[1], b1) & (KV_TAB - 1))] = ids
            sp[1] = b1
            self.dirty = True

    def release(self, row):
        """The row has left: its blocks go back to the pool, its entries to the scratch block."""
        sp = self.span[row]
        if sp is None:
            return
        for bi in range(sp[0], sp[1]):
            self.free.append(int(self.tab_h[row, bi & (KV_TAB - 1)]))
        self.tab_h[row, :] = 0
        self.span[row] = None
        self.dirty = True

    def flush(self):
        """Host mirror -> device table, on the current stream (ordered between the loop's graph replays)."""
        if self.dirty:
            self.tab.copy_(torch.from_numpy(self.tab_h))
            self.dirty = False

    def phys(self, rows, pos):
        """(block, offset) numpy arrays of positions `pos` of table rows `rows` (numpy int arrays of equal length)."""
        import numpy as np
        rows, pos = np.asarray(rows, dtype=np.int64), np.asarray(pos, dtype=np.int64)
        return self.tab_h[rows, (pos >> self.shift) & (KV_TAB - 1)].astype(np.int64), pos & (self.bs - 1)

    def used_blocks(self):
        return self.blocks - 1 - len(self.free)


class GPTEngine:
    def __init__(self, W: dict, layers: int, model_dim: int, heads: int, dtype=torch.bfloat16, device="cuda",
                 start_mel_token=8192, stop_mel_token=8193):
        assert model_dim == heads * 64, "kernels are specialised for head_dim 64"
        self.L, self.D, self.H = layers, model_dim, heads
        self.dtype, self.device = dtype, torch.device(device)
        self.start_mel, self.stop_mel = start_mel_token, stop_mel_token
        dev = self.device

        def f32(k):
            return W[k].detach().to(dev, torch.float32).contiguous()

        def packed(w_kn):
            return nat.pack_weight(w_kn.detach().to(dev, dtype).contiguous())

        # Decode-step structure:
        #   "fold" (default for bf16 / f16): LayerNorm folded into the consuming GEMM -- 5 launches per block.  The QKV and FC
        #            GEMMs multiply the RAW residual rows (a T-typed packed copy `hb` of h) by gamma . W and apply the row
        #            statistics, which they compute themselves on the matrix pipe, in their epilogue (csrc/gemm_skinny.hip);
        #            the out-projection and FC2 run without split-K and add into h in their epilogue (one owner per element,
        #            deterministic), storing hb beside it.  One [LayerNorm + LayerNorm] launch per token is left (ln_f, final_norm).
        #   "launch": every [residual-reduce + LayerNorm] is a launch of its own behind a split-K GEMM -- 7 launches per block.
        #            fp32 parity mode and engines with runtime LoRA adapters (whose (x A) B term rides on the reduce launch).
        # Forms measured and removed in rounds 1-3: LayerNorm in the consumer GEMM's PROLOGUE, LN rows produced by extra
        # workgroups of the consumer's launch, a reducer tail inside the split-K launches.
        self.decode_mode = os.environ.get("ITTS_DECODE_MODE", "fold" if dtype != torch.float32 else "launch")
        if self.decode_mode not in ("fold", "launch"):
            raise ValueError("ITTS_DECODE_MODE must be 'fold' or 'launch'")
        if dtype == torch.float32:
            self.decode_mode = "launch"
        # launch geometry of the two GEMMs that run without split-K in "fold" mode (rows per workgroup, 16-wave workgroups)
        self.fold_rows = [int(v) for v in os.environ.get("ITTS_FOLD_ROWS", "16,16").split(",")]   # out-projection, FC2
        self.fold_rows_consumers = int(os.environ.get("ITTS_FOLD_ROWS_C", "32"))   # QKV' / FC' rows per workgroup when a step has > 32 rows
        self.fold_wide = os.environ.get("ITTS_FOLD_WIDE", "0") == "1"   # measured equal (836.7 / 840.6 us per token)
        # T-typed activations of the decode step (xn, attention output, MLP hidden) live in the packed fragment layout
        # (include/indextts_hip.h): the GEMMs read them as contiguous 1-KiB blocks.  ITTS_PACKED_ACT=0: row-major (same bits).
        self.pa = os.environ.get("ITTS_PACKED_ACT", "1") == "1"
        # beam search: KV rows follow their beams through a row TABLE read by the attention kernel ("table", no cache bytes
        # move) or by permuting the cache rows in place ("copy": itts_beam_reorder_kv, the reference form of
        # GPT2InferenceModel._reorder_cache, model.py:207-218; 1 ms per token at 32 x 3 rows)
        # num_beams = 1: the KV cache is PAGED (block pool + per-row block table, class PagedKV): rows hold only the blocks their
        # own window needs, and continuous batching (decode_refill) hands the blocks of a row that has stopped to the utterance
        # that takes its slot -- the loop runs for as long as the queue lasts inside a fixed pool.  ITTS_PAGED_KV=0: contiguous
        # [rows][H][smax][64] strips (what beam search uses: its per-position row table addresses whole rows).
        self.paged = os.environ.get("ITTS_PAGED_KV", "1") == "1"
        self.kv = None            # PagedKV of the batch in flight (None: contiguous cache)
        self.beam_kv = os.environ.get("ITTS_BEAM_KV", "table")
        self._kv_rows = None   # the table of the beam decode in progress (None outside decode_beam)
        self._shared_prefix = None   # (B, num_beams) after prefill(beams=n): the prompt's K/V exists once per batch element
        self.max_rows_per_launch = 16 if dtype == torch.float32 else 96   # rows one skinny-GEMM launch covers

        # Rows that have emitted their stop token are left out of the decode attention (the sampler pads them with the stop
        # token whatever their logits are).  Their logits -- decode(return_logits=True) -- are then UNDEFINED from the step
        # after their stop on; every other row is untouched (all later stages are per row).  False: compute them anyway.
        self.skip_finished = True
        self._W = W          # kept (by reference) for attach_lora: the engine itself only holds packed copies
        self.lora = False
        self.layers = []

        def folded(ln, wkey, bkey):
            """LN(h; gamma, beta) W + b = rstd (h W' - mean c) + d:  W' = T(gamma . W) packed, c = column sums of the ROUNDED W'
            (so that a constant row cancels exactly), d = beta W + b.  Sums in float64, stored fp32."""
            Wm = W[wkey].detach().to(dev, torch.float64)
            g, bt = ln[0].to(torch.float64), ln[1].to(torch.float64)
            Wr = (g[:, None] * Wm).to(torch.float32).to(dtype)
            c = Wr.to(torch.float64).sum(0).to(torch.float32).contiguous()
            d = (bt @ Wm + W[bkey].detach().to(dev, torch.float64)).to(torch.float32).contiguous()
            return nat.pack_weight(Wr.contiguous()), c, d
        for i in range(layers):
            p = f"gpt.h.{i}."
            d = dict(
                ln1=(f32(p + "ln_1.weight"), f32(p + "ln_1.bias")),
                ln2=(f32(p + "ln_2.weight"), f32(p + "ln_2.bias")),
                w_qkv=packed(W[p + "attn.c_attn.weight"]), b_qkv=f32(p + "attn.c_attn.bias"),
                w_o=packed(W[p + "attn.c_proj.weight"]), b_o=f32(p + "attn.c_proj.bias"),
                w_fc=packed(W[p + "mlp.c_fc.weight"]), b_fc=f32(p + "mlp.c_fc.bias"),
                w_pr=packed(W[p + "mlp.c_proj.weight"]), b_pr=f32(p + "mlp.c_proj.bias"),
            )
            if self.decode_mode == "fold":
                d["wf_qkv"], d["c_qkv"], d["d_qkv"] = folded(d["ln1"], p + "attn.c_attn.weight", p + "attn.c_attn.bias")
                d["wf_fc"], d["c_fc"], d["d_fc"] = folded(d["ln2"], p + "mlp.c_fc.weight", p + "mlp.c_fc.bias")
            self.layers.append(d)
        self.ln_f = (f32("gpt.ln_f.weight"), f32("gpt.ln_f.bias"))
        self.final_norm = (f32("final_norm.weight"), f32("final_norm.bias"))
        self.V = W["mel_head.weight"].shape[0]
        self.w_head = packed(W["mel_head.weight"].t())
        self.b_head = f32("mel_head.bias")
        self.mel_emb = f32("mel_embedding.weight")
        self.mel_pos = f32("mel_pos_embedding.emb.weight")
        self.text_emb = f32("text_embedding.weight")
        self.text_pos = f32("text_pos_embedding.emb.weight")
        self.extra_ids = torch.tensor([1, start_mel_token], dtype=torch.int32, device=dev)  # fake prefix ids (model.py:658-667)
        self._cap_b = self._cap_s = 0
        self._kv_pool = None      # the PagedKV object kept across batches (re-made when it has to grow)
        self._graphs = {}
        # split-K of the two N=1280 GEMMs of a block (80 column tiles): 3 -> 80 x 3 = 240 workgroups of one tile; 6 -> 40 x 6 = 240
        # workgroups of TWO tiles, whose waves fetch each activation fragment once for both (a third less load traffic per CU)
        self.KSPLIT = int(os.environ.get("ITTS_KSPLIT", "3"))
        self.force_eager = False  # measurement aid: launch every kernel eagerly
        self.share_prefix = os.environ.get("ITTS_SHARE_PREFIX", "1") != "0"   # prefill(shared_rows=C): compute the shared rows once
        self.share_kv_reads = os.environ.get("ITTS_SHARE_KV_READS", "1") != "0"   # ... and let the decode attention read them from row 0
        self.steps_per_graph = int(os.environ.get("ITTS_STEPS_PER_GRAPH", "1"))  # decode tokens per CUDA-graph replay; measured 1 > 2 > 4 > 8 (1297 / 1319 / 1342 / 1368 us per token)
        self._sink = torch.zeros(4, dtype=torch.int32, device=dev)
        self.weight_bytes = sum(t.numel() * t.element_size() for l in self.layers for t in
                                (l["w_qkv"], l["w_o"], l["w_fc"], l["w_pr"])) + self.w_head.numel()

    # ------------------------------------------------------------------------------------------------ runtime LoRA
    def attach_lora(self, adapters: dict, scaling: float):
        """Unmerged LoRA adapters at run time (the reference always merges them before saving, train.py:802-832; this is the
        north_star's "LoRA A.B fused into the output projection").  `adapters`: {"gpt.h.{i}.attn.c_proj": (A, B), ...} with
        the peft tensors of a Conv1D target (fan_in_fan_out=True): A = lora_A.weight [r, in], B = lora_B.weight [out, r];
        y = x W + (x A^T) B^T * scaling, scaling = lora_alpha / r (train.py:555-563).
          * output projections (attn.c_proj, mlp.c_proj): stay UNMERGED in the decode loop -- A^T * scaling rides along as
            16*ceil(r/16) extra output columns of the packed weight (the split-K GEMM produces x A next to x W), and the
            [residual-reduce + LayerNorm] launch that follows adds (x A) B^T: no extra launch, no merged weight copy;
          * attn.c_attn / mlp.c_fc adapters, and every adapter in the large-M passes (prefill, latent), are merged into
            packed copies here (their GEMMs have no reduce stage to carry the correction)."""
        dev, T, D = self.device, self.dtype, self.D
        W = self._W

        def f32(k):
            return W[k].detach().to(dev, torch.float32)

        def packed(w):
            return nat.pack_weight(w.to(T).contiguous())

        self.detach_lora()   # every attach starts from the base weights: nothing of an earlier adapter set survives
        if not adapters:     # attach_lora(None) = detach
            return
        for i, l in enumerate(self.layers):
            p = f"gpt.h.{i}."
            for name, wkey in (("attn.c_attn", "w_qkv"), ("mlp.c_fc", "w_fc")):
                if p + name in adapters:
                    A, Bm = (t.detach().to(dev, torch.float32) for t in adapters[p + name])
                    l[wkey + "_base"] = l[wkey]
                    l[wkey] = packed(f32(p + name + ".weight") + (A.t() @ Bm.t()) * scaling)
            for name, wkey, tag in (("attn.c_proj", "w_o", "o"), ("mlp.c_proj", "w_pr", "pr")):
                if p + name not in adapters:
                    continue
                A, Bm = (t.detach().to(dev, torch.float32) for t in adapters[p + name])
                r = A.shape[0]
                if r > 64:
                    raise ValueError("runtime LoRA supports rank <= 64")
                rp = (r + 15) // 16 * 16
                base = f32(p + name + ".weight")                                  # [K, D]
                ext = torch.zeros(base.shape[0], D + rp, dtype=torch.float32, device=dev)
                ext[:, :D] = base
                ext[:, D:D + r] = A.t() * scaling
                l[wkey + "_lora"] = packed(ext)                                   # decode: [W | A^T s], N = D + rp
                l["lora_b_" + tag] = Bm.t().contiguous()                          # [r, D] fp32, applied by ln_reduce
                l["lora_n_" + tag] = D + rp
                l[wkey + "_merged"] = packed(base + (A.t() @ Bm.t()) * scaling)   # large-M passes
        self.lora = True
        self._graphs.clear()
        if hasattr(self, "slab") and self.slab.shape[2] < D + 64:
            self._cap_b = self._cap_s = 0   # the slabs need room for the extra columns: reallocate on the next prefill

    def detach_lora(self):
        """Back to the base weights: drops every adapter-derived entry (extra-column projections, merged copies, B factors)
        and the graphs captured over them.  Only this engine changes: a fork holds its own layer dicts."""
        for l in self.layers:
            for k in [k for k in l if k.endswith("_lora") or k.endswith("_merged") or k.startswith("lora_")]:
                del l[k]
            for wkey in ("w_qkv", "w_fc"):
                if wkey + "_base" in l:
                    l[wkey] = l.pop(wkey + "_base")
        self.lora = False
        self._graphs.clear()

    def fork(self) -> "GPTEngine":
        """A second engine over the SAME packed weights (read-only, shared tensors) with its own KV cache, scratch buffers,
        loop state, captured graphs and per-layer dicts: what a concurrent request needs (infer.RequestPool).  A fork keeps
        the adapter state it was forked with; attach_lora / detach_lora on one engine never changes another."""
        import copy
        e = copy.copy(self)
        e.layers = [dict(l) for l in self.layers]
        e._cap_b = e._cap_s = 0
        e.kv = None
        e._kv_pool = None
        e._graphs = {}
        e._beam_cap = (0, 0, 0)
        e._kv_rows = None
        e._sink = torch.zeros(4, dtype=torch.int32, device=self.device)
        return e

    # ------------------------------------------------------------------------------------------------ buffers
    def _ensure(self, B: int, smax: int, paged=False, blocks=0, bs=16):
        """Buffers for B rows.  Contiguous cache: smax positions per row.  Paged cache: a pool of `blocks` blocks of `bs`
        positions (smax is then only the nominal window, nothing is sized by it)."""
        dev, T = self.device, self.dtype
        if B > self._cap_b:
            B = max(B, self._cap_b)
            self.h = torch.zeros(B, self.D, dtype=torch.float32, device=dev)
            Bp = nat.packed_rows(B)   # the packed layout works in 16-row tiles
            self.q = torch.zeros(B, self.D, dtype=T, device=dev)
            self.a = torch.zeros(Bp, self.D, dtype=T, device=dev)
            self.f = torch.zeros(Bp, 4 * self.D, dtype=T, device=dev)
            self.xn = torch.zeros(Bp, self.D, dtype=T, device=dev)
            self.hb = torch.zeros(Bp, self.D, dtype=T, device=dev)   # "fold" mode: T-typed packed copy of the residual rows
            self.slab = torch.zeros(self.KSPLIT, B, self.D + 64, dtype=torch.float32, device=dev)   # + runtime-LoRA columns
            self.logits = torch.zeros(B, self.V, dtype=torch.float32, device=dev)
            self.tokens = torch.zeros(B, dtype=torch.int32, device=dev)
            self.finished = torch.zeros(B, dtype=torch.int32, device=dev)
            self.pad = torch.zeros(B, dtype=torch.int32, device=dev)
            self.force_stop = torch.full((B,), -1, dtype=torch.int32, device=dev)
            self.row_step0 = torch.zeros(B, dtype=torch.int32, device=dev)   # loop step at which each row started (decode_refill)
            self.kv_share = torch.zeros(1, dtype=torch.int32, device=dev)    # (p0 << 8) | C: rows' first C keys == row 0's at p0 (itts_attn_decode)
            self.state = torch.zeros(8, dtype=torch.int32, device=dev)
            self.history = torch.zeros(B, 2048, dtype=torch.int32, device=dev)
            self._cap_b = B
            self._cap_s = 0          # the contiguous cache follows the row buffers
            self._kv_pool = None
            self._graphs.clear()
        if paged:
            kv = self._kv_pool
            if kv is None or kv.rows < self._cap_b or kv.blocks < blocks or kv.bs != bs:
                kv = None
                self.kv = self._kv_pool = None          # release the old pool before the new one is allocated
                kv = self._kv_pool = PagedKV(self.L, self.H, self._cap_b, max(blocks, 2), bs, T, dev)
                self._graphs.clear()
            else:
                kv.reset()
            self.kv = kv
            self.kc, self.vc = kv.kc, kv.vc
            if self._cap_s:                              # a contiguous cache of an earlier (beam) batch: its graphs go with it
                self._cap_s = 0
                self._kc_rows = self._vc_rows = None
                self._graphs.clear()
            return
        self.kv = None
        if smax > self._cap_s:
            smax = (smax + 63) // 64 * 64
            if smax > 16384:
                raise ValueError(f"context {smax} exceeds the decode-attention limit of 16384 cache positions")
            self._kv_pool = None
            self._kc_rows = torch.zeros(self.L, self._cap_b, self.H, smax, 64, dtype=T, device=dev)
            self._vc_rows = torch.zeros(self.L, self._cap_b, self.H, smax, 64, dtype=T, device=dev)
            self._cap_s = smax
            self._graphs.clear()
        self.kc, self.vc = self._kc_rows, self._vc_rows

    def _kvargs(self):
        """Keyword arguments that tell a kernel where position j of a cache row lives."""
        return dict(kv_tab=self.kv.tab, kv_bs=self.kv.bs) if self.kv is not None else {}

    def dense_kv(self, rows: int, S: int):
        """(K, V) T [L, rows, H, S, 64] of cache positions [0, S): a dense copy whatever the cache layout (tests, tools)."""
        if self.kv is None:
            return self.kc[:, :rows, :, :S].clone(), self.vc[:, :rows, :, :S].clone()
        import numpy as np
        r, p = np.repeat(np.arange(rows), S), np.tile(np.arange(S), rows)
        blk, off = (torch.from_numpy(a).to(self.device) for a in self.kv.phys(r, p))
        k = self.kc[:, blk, :, off].view(rows, S, self.L, self.H, 64).permute(2, 0, 3, 1, 4).contiguous()
        v = self.vc[:, blk, :, off].view(rows, S, self.L, self.H, 64).permute(2, 0, 3, 1, 4).contiguous()
        return k, v

    # ------------------------------------------------------------------------------------------------ big-M passes
    def _proj_ksplit(self, M):
        """Split-K factor for the two N = D projections of a big-M pass: their 128 x 128 output tiles number M/128 x 10 -- 160 at the
        prefill's ~2 000 rows.  2-4 slices fill the chip (the factor is chosen below); the slabs are summed by the LayerNorm launch that
        follows (itts_ln_reduce: residual + bias + slabs, then LN), which replaces the GEMM's residual epilogue AND the LayerNorm
        launch.  1 where the tiles alone already fill the slots (the latent pass)."""
        if self.dtype == torch.float32 or self.D % 256 or os.environ.get("ITTS_PREFILL_KSPLIT", "1") == "0":
            return 1
        tiles = ((M + 127) // 128) * (self.D // 128)
        force = os.environ.get("ITTS_PREFILL_KS")          # measurement aid (tools/): a fixed factor
        if force:
            return max(1, int(force))
        if tiles > 270:
            return 1
        # a CU's share of the work, in whole-K tile units: ceil(tiles x ks / 256 CUs) slices of 1 / ks each (the kernel is priced by the
        # CU's load path, so two co-resident slices take twice one slice).  Config 3's prefill (M ~ 1 400: 110 tiles): ks = 2 -> 0.5
        # against 0.67 for ks = 3 (74 CUs would hold two slices) and 1.0 unsplit: 4.2 -> 3.8 ms; M ~ 2 000 (160 tiles): ks = 3 -> 0.67
        cus = 256
        cost = lambda ks: -(-tiles * ks // cus) / ks                                  # noqa: E731
        rule = 3 if tiles * 3 <= 540 else 2 if tiles * 2 <= 540 else 1               # round 3's choice (512 workgroup slots)
        best = min((1, 2, 3), key=lambda ks: (cost(ks), ks))
        return best if cost(best) < 0.9 * cost(rule) else rule

    def _big_m_layers(self, h, attn):
        """The 24 blocks over packed rows h fp32 [M, D] (in place): LayerNorm -> QKV -> attn(i, qkv, att) -> out-projection ->
        LayerNorm -> FC -> FC2.  With few rows the two N = D projections run split-K into slabs and the NEXT LayerNorm launch folds
        them into h (see _proj_ksplit); returns h with every block applied."""
        T, D, dev = self.dtype, self.D, self.device
        M = h.shape[0]
        xn = torch.empty(M, D, dtype=T, device=dev)
        qkv = torch.empty(M, 3 * D, dtype=T, device=dev)
        att = torch.empty(M, D, dtype=T, device=dev)
        ff = torch.empty(M, 4 * D, dtype=T, device=dev)
        ks = self._proj_ksplit(M)
        slab = torch.empty(ks, M, D, dtype=torch.float32, device=dev) if ks > 1 else None
        pending = None                       # bias of the projection whose slabs the next LayerNorm launch has to fold in
        for i, l in enumerate(self.layers):
            if pending is None:
                nat.layernorm(h, l["ln1"][0], l["ln1"][1], xn)
            else:
                nat.ln_reduce(h, l["ln1"][0], l["ln1"][1], xn, slab=slab, nslab=ks, bias=pending)
            nat.gemm_conv(T, 1, M, M, D, 3 * D, l["w_qkv"], xn, qkv, bias=l["b_qkv"])
            attn(i, qkv, att)
            w_o, w_pr = l.get("w_o_merged", l["w_o"]), l.get("w_pr_merged", l["w_pr"])
            if ks > 1:
                nat.gemm_conv(T, 1, M, M, D, D, w_o, att, slab, y_f32=True, ksplit=ks)
                nat.ln_reduce(h, l["ln2"][0], l["ln2"][1], xn, slab=slab, nslab=ks, bias=l["b_o"])
            else:
                nat.gemm_conv(T, 1, M, M, D, D, w_o, att, h, bias=l["b_o"], y_f32=True, resid=h)
                nat.layernorm(h, l["ln2"][0], l["ln2"][1], xn)
            nat.gemm_conv(T, 1, M, M, D, 4 * D, l["w_fc"], xn, ff, bias=l["b_fc"], act=1)
            if ks > 1:
                nat.gemm_conv(T, 1, M, M, 4 * D, D, w_pr, ff, slab, y_f32=True, ksplit=ks)
                pending = l["b_pr"]
            else:
                nat.gemm_conv(T, 1, M, M, 4 * D, D, w_pr, ff, h, bias=l["b_pr"], y_f32=True, resid=h)
        if pending is not None:              # the last block's FC2 slabs: fold them in (the LayerNorm output is not used)
            nat.ln_reduce(h, self.ln_f[0], self.ln_f[1], xn, slab=slab, nslab=ks, bias=pending)
        return h

    def _blocks_full(self, h, B, S, pad, use_cache, row_off=None, cache_shift=None):
        """All transformer blocks over h fp32 [M, D] (in place).  Padded form: M = B*S rows, pad int32 [B] (left padding)
        or None.  Packed form (row_off int32 [B+1] on device): only real rows exist, batch element b owns rows
        [row_off[b], row_off[b+1]), S is the longest element, cache row = cache_shift[b] + local row."""
        H = self.H
        kva = self._kvargs() if use_cache else {}

        def attn(i, qkv, att):
            kc, vc = (self.kc[i], self.vc[i]) if use_cache else (None, None)
            if row_off is None:
                nat.attn_prefill(qkv, att, kc, vc, pad, B, S, H, self._cap_s)
            else:
                nat.attn_prefill_packed(qkv, att, kc, vc, row_off, cache_shift, B, S, H, self._cap_s, **kva)
        return self._big_m_layers(h, attn)

    def _head(self, h_rows, B):
        """ln_f -> final_norm -> mel_head on fp32 rows."""
        nat.ln_reduce(h_rows, self.ln_f[0], self.ln_f[1], self.xn, w2=self.final_norm[0], b2=self.final_norm[1], y_packed=self.pa)
        nat.gemm_skinny(self.dtype, B, self.V, self.D, self.w_head, self.b_head, x=self.xn, epi=nat.EPI_STORE_F32,
                        yf=self.logits, x_packed=self.pa)

    def _prefill_shared(self, emb, pad_h, S, C):
        """The packed prefill pass for a batch whose elements all begin with the SAME C rows (one prompt's conditioning
        latents: no position embedding is added to them and they see only themselves, so their hidden states, keys and values
        are the same in every element).  Those C rows go through the blocks ONCE; every element contributes only its own rows
        (text + start token), whose attention reads the shared block's keys / values out of the same qkv buffer
        (itts_attn_prefill_shared: key tiles cut from sequence position 0 as in the un-shared pass, same bits per row).  The
        kernel appends every element's own rows to its cache row; the shared block's cache rows are copied to the other
        elements afterwards (itts_kv_share_rows: all layers in one launch).  Returns the hidden states of the elements' last rows."""
        import numpy as np
        T, D, H, dev = self.dtype, self.D, self.H, self.device
        B = emb.shape[0]
        own = [S - p - C for p in pad_h]                     # rows of each element behind the shared block
        if min(own) < 1:
            raise ValueError("prefill(shared_rows): an element has no row of its own behind the shared block")
        off = np.concatenate([[0, C], C + np.cumsum(own)]).astype(np.int64)
        M = int(off[-1])
        idx = np.concatenate([pad_h[0] + np.arange(C)] + [b * S + pad_h[b] + C + np.arange(own[b]) for b in range(B)])
        meta = torch.from_numpy(np.concatenate([
            idx, off, [0] + [C] * B, np.zeros(B + 1, np.int64), [0] + list(range(B)), [pad_h[0]] + [p + C for p in pad_h],
            off[2:] - 1]).astype(np.int64)).to(dev)                                                         # one upload
        E = B + 1
        o = 0

        def take(n, as32=True):
            nonlocal o
            t = meta[o:o + n]
            o += n
            return t.to(torch.int32) if as32 else t
        i_rows, row_off = take(M, False), take(E + 1)
        pre_len, pre_row0, w_row, w_pos0 = take(E), take(E), take(E), take(E)
        last_rows = take(B, False)
        h = emb.view(B * S, D)[i_rows]
        Smax = max(C, max(own))
        kva = self._kvargs()
        h = self._big_m_layers(h, lambda i, qkv, att: nat.attn_prefill_shared(
            qkv, att, self.kc[i], self.vc[i], row_off, pre_len, pre_row0, w_row, w_pos0, E, Smax, H, self._cap_s, **kva))
        if B > 1:   # the shared block's keys / values: cache row 0 -> the same sequence positions of every other cache row
            nat.kv_share_rows(self.kc, self.vc, B, H, C, pad_h[0], self.pad, self._cap_s, **self._kvargs())
        return h[last_rows].contiguous()

    def prefill(self, prefix_emb: torch.Tensor, pad: torch.Tensor, max_new: int, beams: int = 1, shared_rows: int = 0, paged=None,
                slots_window: int = 0):
        """prefix_emb fp32 [B,P,D] (left-padded with zeros), pad int [B].  Runs prefix + start token (mel position 0,
        model.py:152-162), fills the KV cache rows [pad_b, P] of every element, leaves logits of the last position in
        self.logits.  The left-padding rows are never computed: the real rows are packed (one gather), the GEMMs run over
        sum(P + 1 - pad_b) rows instead of B*(P+1), and the attention kernel writes each element's keys at its padded
        cache position, so the decode loop sees the reference's left-padded cache layout.
        shared_rows = C > 0: the caller promises that the first C real rows of every element are THE SAME rows (one prompt's
        conditioning latents, model.py:606-667 with a [1, C, D] conditioning tensor); they are then computed once for the
        batch (_prefill_shared: same logits and cache contents, a third fewer GEMM rows at config 3's shape).
        beams > 1 (beam search with the row table): HF expands every row to `beams` identical rows BEFORE the first forward;
        here the prompt is computed and cached ONCE per batch element (cache rows 0..B-1) and decode_beam() points the table
        entries of all its beams at that copy -- a third of the prefill work and prompt cache at 3 beams, identical values
        (the expanded rows are bit-wise copies).  The engine then holds B*beams rows: logits and pad are expanded here.
        paged (default: the engine's setting for beams == 1, never with beams > 1): the cache is a block pool behind a block
        table (class PagedKV).  Every row gets the blocks for its own window [pad_b, S + max_new]; a caller that goes on with
        decode_beam() over expanded rows passes paged=False (beam search addresses whole contiguous rows).  slots_window > 0
        (decode_refill's pool): size the pool for B rows of up to that many live positions each instead of this batch's."""
        B, P, D = prefix_emb.shape
        S = P + 1
        beams = int(beams)
        if beams > 1 and self.beam_kv != "table":
            raise ValueError("prefill(beams>1) needs the KV row table (beam_kv='table')")
        pad_h = [int(v) for v in torch.as_tensor(pad).tolist()]
        use_pages = (self.paged if paged is None else bool(paged)) and beams == 1
        if use_pages:
            window = max(S + max_new + 1 - min(pad_h), int(slots_window))
            bs = 16 if window <= (KV_TAB - 2) * 16 else 32 if window <= (KV_TAB - 2) * 32 else 64
            if window > (KV_TAB - 2) * 64:
                use_pages = False      # a window no block table of 64 entries covers: contiguous rows
        if use_pages:
            span = lambda lo, hi: ((hi - 1) // bs) - (lo // bs) + 1   # noqa: E731  blocks that cover positions [lo, hi)
            need = sum(span(p, S + max_new + 1) for p in pad_h)
            if slots_window:
                need = max(need, B * (int(slots_window) // bs + 2))
            self._ensure(B, S + max_new + 1, paged=True, blocks=need + 1, bs=bs)
            for b, p in enumerate(pad_h):
                self.kv.cover(b, p, S + max_new + 1)
            self.kv.flush()
        else:
            self._ensure(B * beams, S + max_new + 1)
        dev = self.device
        start = self.mel_emb[self.start_mel] + self.mel_pos[0]
        emb = torch.cat([prefix_emb.to(dev, torch.float32), start.expand(B, 1, D)], dim=1).contiguous()
        self._pad_host = pad_h             # latent_mel_rows() finds the prompt's K/V in the cache through it
        self.pad[:B] = torch.tensor(pad_h, dtype=torch.int32).to(dev)
        self.kv_share.zero_()
        if shared_rows and B > 1 and self.share_prefix:
            # every element starts with the same `shared_rows` rows (the caller's promise: one prompt's conditioning latents)
            self._head(self._prefill_shared(emb, pad_h, S, int(shared_rows)), B)
            if beams == 1 and int(shared_rows) <= 255 and self.share_kv_reads:
                # the decode attention reads those rows' keys / values from cache row 0 for every row (same bytes, one copy in L2)
                self.kv_share.fill_((pad_h[0] << 8) | int(shared_rows))
        else:
            lens = [S - p for p in pad_h]
            off = [0]
            for n in lens:
                off.append(off[-1] + n)
            idx = torch.cat([torch.arange(b * S + pad_h[b], (b + 1) * S) for b in range(B)])
            meta = torch.tensor(off + [b_off - 1 for b_off in off[1:]], dtype=torch.int32).to(dev)   # row_off | last rows
            row_off, last_rows = meta[: B + 1], meta[B + 1:].long()
            h = emb.view(B * S, D)[idx.to(dev)]
            h = self._blocks_full(h, B, S, None, True, row_off=row_off, cache_shift=self.pad[:B])
            self._head(h[last_rows].contiguous(), B)
        self.state.zero_()                 # step, cache position, finished rows, arrival counter, seed (lo, hi)
        self.state[1] = S - 1
        self._pending_bump = False
        self.finished[:B] = 0
        self.row_step0.zero_()
        self.history[:B].zero_()
        self._B, self._S = B, S
        self._shared_prefix = None
        if beams > 1:
            R = B * beams
            self.logits[:R] = self.logits[:B].repeat_interleave(beams, dim=0)
            self.pad[:R] = self.pad[:B].repeat_interleave(beams)
            self._B = R
            self._shared_prefix = (B, beams)
        return self.logits[: self._B]

    def latent(self, emb: torch.Tensor, lengths=None) -> torch.Tensor:
        """Teacher-forced pass (model.py:459-474): emb fp32 [B,S,D] (right-padded rows allowed) ->
        final_norm(ln_f(blocks(emb))) fp32 [B,S,D].  With `lengths` (host ints, real rows per element) only the real rows
        are computed (packed); the padding rows of the result are zero."""
        B, S, D = emb.shape
        if self._cap_b == 0:
            self._ensure(1, 64)
        dev = self.device
        src = emb.to(dev, torch.float32).contiguous().view(B * S, D)
        if lengths is None:
            h = src.clone()
            self._blocks_full(h, B, S, None, False)
            out = torch.empty_like(h)
            nat.layernorm(h, self.ln_f[0], self.ln_f[1], out, self.final_norm[0], self.final_norm[1])
            return out.view(B, S, D)
        lens = [int(n) for n in lengths]
        off = [0]
        for n in lens:
            off.append(off[-1] + n)
        idx = torch.cat([torch.arange(b * S, b * S + lens[b]) for b in range(B)]).to(dev)
        row_off = torch.tensor(off, dtype=torch.int32).to(dev)
        h = src[idx]
        self._blocks_full(h, B, max(lens), None, False, row_off=row_off)
        packed = torch.empty_like(h)
        nat.layernorm(h, self.ln_f[0], self.ln_f[1], packed, self.final_norm[0], self.final_norm[1])
        out = torch.zeros(B * S, D, dtype=torch.float32, device=dev)
        out[idx] = packed
        return out.view(B, S, D)

    def latent_mel_rows(self, mel_emb: torch.Tensor, m_lens, cache_rows=None) -> torch.Tensor:
        """The teacher-forced pass (model.py:459-474, :548-597) over the MEL rows only, for the batch whose prompt the last
        prefill() cached.  In cond | text | mel the causal mask lets no prompt position see a mel position, so the prompt's
        keys and values in every layer are exactly what prefill() computed for the decode loop -- same kernels, same inputs,
        bit for bit -- and they are still in the KV cache at positions [pad_b, P) (the decode loop only appends behind them).
        Only the mel rows (start, codes, stop: ~60 % of the sequence) go through the GEMMs, LayerNorms and attention queries;
        the flash-attention kernel reads the prompt's K / V tiles from the cache rows and the mel rows' from this pass's qkv,
        with its key tiles cut from sequence position 0 as in latent().  Same bits as latent() on the whole sequence
        (tests/test_engines_gpu.py::test_latent_pass_reuses_the_cached_prompt).
        mel_emb fp32 [sum(m_lens), D]: the mel segments' embeddings, elements in prefill order; cache_rows: the cache row that
        holds element b's prompt (default b; b * num_beams after a beam prefill that copied rows).
        Returns final_norm(ln_f(hidden)) fp32 [sum(m_lens), D]."""
        import numpy as np
        T, D, H, dev = self.dtype, self.D, self.H, self.device
        B, P = len(m_lens), self._S - 1
        cache_rows = list(range(B)) if cache_rows is None else [int(r) for r in cache_rows]
        if len(cache_rows) != B or max(cache_rows) >= len(self._pad_host):
            raise ValueError("latent_mel_rows(): the batch differs from the one prefill() cached")
        pads = [self._pad_host[r] for r in cache_rows]   # (a prefill of expanded beam rows lists the padding per cache row)
        pl = [P - p for p in pads]                        # real prompt rows per element
        m = [int(v) for v in m_lens]
        off = np.concatenate([[0], np.cumsum(m)])
        Mm = int(off[-1])
        if mel_emb.shape[0] != Mm:
            raise ValueError("latent_mel_rows(): mel_emb does not hold sum(m_lens) rows")
        meta = torch.from_numpy(np.concatenate([off, pl, cache_rows, pads]).astype(np.int32)).to(dev)   # one upload
        row_off, pre_len, pre_row, pre_pos0 = meta[: B + 1], meta[B + 1:2 * B + 1], meta[2 * B + 1:3 * B + 1], meta[3 * B + 1:]
        h = mel_emb.to(dev, torch.float32).contiguous()
        kva = self._kvargs()
        # the prompt's keys / values straight from the decode cache (itts_attn_prefill_prefix), the mel rows' from qkv
        h = self._big_m_layers(h, lambda i, qkv, att: nat.attn_prefill_prefix(
            qkv, att, self.kc[i], self.vc[i], row_off, pre_len, pre_row, pre_pos0, B, max(m), H, self._cap_s, **kva))
        out = torch.empty_like(h)
        nat.layernorm(h, self.ln_f[0], self.ln_f[1], out, self.final_norm[0], self.final_norm[1])
        return out

    # ------------------------------------------------------------------------------------------------ decode loop
    def _sample(self, B, sp, dbg=None):
        """Token selection for all rows.  The loop state (step counter, cache position) is NOT advanced here: the next
        transformer step does it in its first LayerNorm launch (a launch that reads neither word), which takes a
        device-wide fence and a returning atomic per row out of the sampling kernel."""
        nat.sample(self.logits[:B], self.tokens, self.history, self.finished, self.state, self.extra_ids, self.force_stop,
                   sp["repetition_penalty"], sp["temperature"], sp["top_k"], sp["top_p"], sp["do_sample"], sp["seed"],
                   self.stop_mel, dbg, no_advance=True, row_step0=self.row_step0)
        self._pending_bump = True

    def _fold_now(self, B):
        return self.decode_mode == "fold" and not self.lora

    def _step_transformer(self, B, bump=None):
        """(bump: advance step counter / cache position inside this step's first launches; None = "a _sample call is
        waiting for it", which is what the token loop wants; the beam step kernel advances the state itself.)
        Transformer part of one cached decode step (model.py:163-193): embed token k at mel position k+1, 24 blocks,
        head.  "fold" form, 5 launches per block: QKV' (LayerNorm folded in, + K/V append) -> attention -> out-projection
        (+ residual update, T copy) -> FC' (folded, + gelu) -> FC2 (+ residual update, T copy); the loop state is advanced by
        launches that do not read the word they bump (embed_step: cache position; the first QKV': step counter).
        "launch" form, 7 per block: split-K slabs + [residual-reduce + LayerNorm] launches (itts_ln_reduce)."""
        T, D, H, KS = self.dtype, self.D, self.H, self.KSPLIT
        step, pos = self.state[0:1], self.state[1:2]
        h, xn, pa = self.h[:B], self.xn, self.pa   # xn / a / f: whole buffers (packed layout is addressed from the base)
        if bump is None:
            bump = getattr(self, "_pending_bump", False)
        self._pending_bump = False
        rs0 = self.row_step0 if self._kv_rows is None else None
        kva = self._kvargs()
        if self._fold_now(B):
            hb = self.hb
            # mel position of token k is k + 1 (model.py:163-167); with a pending bump state[0] still holds k - 1
            nat.embed_step(self.tokens, self.mel_emb, self.mel_pos, step, 2 if bump else 1, h, row_step0=rs0, h_packed=hb,
                           bump=pos if bump else None)
            r_o, r_p = self.fold_rows
            # more than 32 rows (beam search: 32 x 3): a workgroup that covers ALL rows moves 246 KB of activations through its
            # CU's load path per 1280-deep K; 32 rows per workgroup (the row tiles dealt to grid.z, 3-4 column tiles each so that
            # the grid still fits the chip) moves a third of that
            r_c = self.fold_rows_consumers if B > 32 else 0
            if B > 32:
                r_o, r_p = max(r_o, 32), max(r_p, 32)
            for i, l in enumerate(self.layers):
                nat.gemm_skinny(T, B, 3 * D, D, l["wf_qkv"], l["d_qkv"], x=hb, epi=nat.EPI_QKV_CACHE, y=self.q, kcache=self.kc[i],
                                vcache=self.vc[i], pos=pos, heads=H, smax=self._cap_s, x_packed=True, ln_c=l["c_qkv"],
                                bump=step if (bump and i == 0) else None, rows_per_wg=r_c, **kva)
                nat.attn_decode(self.q, self.kc[i], self.vc[i], self.a, self.pad, pos, B, H, self._cap_s, out_packed=pa,
                                kv_rows=self._kv_rows, kv_step=step if self._kv_rows is not None else None,
                                skip_rows=self.finished if self._kv_rows is None and self.skip_finished else None,
                                kv_share=self.kv_share if self._kv_rows is None else None, **kva)
                nat.gemm_skinny(T, B, D, D, l["w_o"], l["b_o"], x=self.a, epi=nat.EPI_RESID_F32, yf=h, y=hb, x_packed=pa,
                                y_packed=True, rows_per_wg=r_o, wide_wg=self.fold_wide)
                nat.gemm_skinny(T, B, 4 * D, D, l["wf_fc"], l["d_fc"], x=hb, epi=nat.EPI_GELU_STORE, y=self.f, x_packed=True,
                                y_packed=pa, ln_c=l["c_fc"], rows_per_wg=r_c)
                nat.gemm_skinny(T, B, D, 4 * D, l["w_pr"], l["b_pr"], x=self.f, epi=nat.EPI_RESID_F32, yf=h, y=hb, x_packed=pa,
                                y_packed=True, rows_per_wg=r_p, wide_wg=self.fold_wide)
            nat.ln_reduce(h, self.ln_f[0], self.ln_f[1], xn, w2=self.final_norm[0], b2=self.final_norm[1], y_packed=pa)
            nat.gemm_skinny(T, B, self.V, D, self.w_head, self.b_head, x=self.xn, epi=nat.EPI_STORE_F32, yf=self.logits, x_packed=pa)
            return
        nat.embed_step(self.tokens, self.mel_emb, self.mel_pos, step, 2 if bump else 1, h, row_step0=rs0)
        nat.ln_reduce(h, self.layers[0]["ln1"][0], self.layers[0]["ln1"][1], xn, state_bump=self.state[0:2] if bump else None,
                      y_packed=pa)
        for i, l in enumerate(self.layers):
            last = i + 1 == self.L
            n_o = l.get("lora_n_o", D)
            n_p = l.get("lora_n_pr", D)
            w_o = l.get("w_o_lora", l["w_o"])
            w_pr = l.get("w_pr_lora", l["w_pr"])
            nat.gemm_skinny(T, B, 3 * D, D, l["w_qkv"], l["b_qkv"], x=xn, epi=nat.EPI_QKV_CACHE, y=self.q, kcache=self.kc[i],
                            vcache=self.vc[i], pos=pos, heads=H, smax=self._cap_s, x_packed=pa, **kva)
            nat.attn_decode(self.q, self.kc[i], self.vc[i], self.a, self.pad, pos, B, H, self._cap_s, out_packed=pa,
                            kv_rows=self._kv_rows, kv_step=step if self._kv_rows is not None else None,
                            skip_rows=self.finished if self._kv_rows is None and self.skip_finished else None,
                            kv_share=self.kv_share if self._kv_rows is None else None, **kva)
            nxt = self.ln_f if last else self.layers[i + 1]["ln1"]
            nxt2 = self.final_norm if last else None
            # out-projection: split-K slabs; with a runtime adapter the GEMM also produces x A in extra columns and the
            # reduce launch adds (x A) B^T
            sl_o = self.slab.view(-1)[: KS * B * n_o].view(KS, B, n_o)
            nat.gemm_skinny(T, B, n_o, D, w_o, None, x=self.a, epi=nat.EPI_SLAB_F32, yf=sl_o, ksplit=KS,
                            x_packed=pa)
            nat.ln_reduce(h, l["ln2"][0], l["ln2"][1], xn, slab=sl_o, nslab=KS, bias=l["b_o"], y_packed=pa,
                          slab_stride=n_o, lora_b=l.get("lora_b_o"))
            nat.gemm_skinny(T, B, 4 * D, D, l["w_fc"], l["b_fc"], x=xn, epi=nat.EPI_GELU_STORE, y=self.f, x_packed=pa, y_packed=pa)
            sl_p = self.slab.view(-1)[: KS * B * n_p].view(KS, B, n_p)
            nat.gemm_skinny(T, B, n_p, 4 * D, w_pr, None, x=self.f, epi=nat.EPI_SLAB_F32, yf=sl_p, ksplit=KS,
                            x_packed=pa)
            if last:
                nat.ln_reduce(h, nxt[0], nxt[1], xn, slab=sl_p, nslab=KS, bias=l["b_pr"], w2=nxt2[0], b2=nxt2[1], y_packed=pa,
                              slab_stride=n_p, lora_b=l.get("lora_b_pr"))
            else:
                nat.ln_reduce(h, nxt[0], nxt[1], xn, slab=sl_p, nslab=KS, bias=l["b_pr"], y_packed=pa, slab_stride=n_p,
                              lora_b=l.get("lora_b_pr"))
        nat.gemm_skinny(T, B, self.V, D, self.w_head, self.b_head, x=self.xn, epi=nat.EPI_STORE_F32, yf=self.logits, x_packed=pa)

    def _poll(self):
        """One host synchronisation of the token loop: the number of finished rows."""
        return int(self.state[2].item())

    def _step_kernels(self, B, sp):
        self._step_transformer(B)
        self._sample(B, sp)

    def gemm_launches_of_step(self, B):
        """Measurement aid (bench.py): ONLY the skinny-GEMM launches of one decode step, with the step's real arguments
        (97 launches: 4 per block + the head).  Returns (GEMM launch count, algorithmic bytes: weights once + activations in
        and out; "fold" mode: the residual rows the two epilogues read and write and their T copy; "launch" mode: the slabs)."""
        T, D, H, KS = self.dtype, self.D, self.H, self.KSPLIT
        pos = self.state[1:2]
        h, xn, pa = self.h[:B], self.xn, self.pa
        es = 4 if T == torch.float32 else 2
        fold = self._fold_now(B)
        nbytes, n = 0, 0
        slab = self.slab.view(-1)[: KS * B * D].view(KS, B, D)
        r_o, r_p = self.fold_rows
        for i, l in enumerate(self.layers):
            if fold:
                hb = self.hb
                nat.gemm_skinny(T, B, 3 * D, D, l["wf_qkv"], l["d_qkv"], x=hb, epi=nat.EPI_QKV_CACHE, y=self.q, kcache=self.kc[i],
                                vcache=self.vc[i], pos=pos, heads=H, smax=self._cap_s, x_packed=True, ln_c=l["c_qkv"], **self._kvargs())
                nat.gemm_skinny(T, B, D, D, l["w_o"], l["b_o"], x=self.a, epi=nat.EPI_RESID_F32, yf=h, y=hb, x_packed=pa,
                                y_packed=True, rows_per_wg=r_o, wide_wg=self.fold_wide)
                nat.gemm_skinny(T, B, 4 * D, D, l["wf_fc"], l["d_fc"], x=hb, epi=nat.EPI_GELU_STORE, y=self.f, x_packed=True,
                                y_packed=pa, ln_c=l["c_fc"])
                nat.gemm_skinny(T, B, D, 4 * D, l["w_pr"], l["b_pr"], x=self.f, epi=nat.EPI_RESID_F32, yf=h, y=hb, x_packed=pa,
                                y_packed=True, rows_per_wg=r_p, wide_wg=self.fold_wide)
                nbytes += 12 * D * D * es + B * D * es * (1 + 1 + 1 + 4) + B * es * (3 * D + 4 * D) + 2 * (2 * B * D * 4 + B * D * es)
            else:
                nat.gemm_skinny(T, B, 3 * D, D, l["w_qkv"], l["b_qkv"], x=xn, epi=nat.EPI_QKV_CACHE, y=self.q, kcache=self.kc[i],
                                vcache=self.vc[i], pos=pos, heads=H, smax=self._cap_s, x_packed=pa, **self._kvargs())
                nat.gemm_skinny(T, B, D, D, l["w_o"], None, x=self.a, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=KS, x_packed=pa)
                nat.gemm_skinny(T, B, 4 * D, D, l["w_fc"], l["b_fc"], x=xn, epi=nat.EPI_GELU_STORE, y=self.f, x_packed=pa, y_packed=pa)
                nat.gemm_skinny(T, B, D, 4 * D, l["w_pr"], None, x=self.f, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=KS, x_packed=pa)
                nbytes += 12 * D * D * es + B * D * es * (1 + 1 + 1 + 4) + B * es * (3 * D + 4 * D) + 2 * KS * B * D * 4
            n += 4
        nat.gemm_skinny(T, B, self.V, D, self.w_head, self.b_head, x=self.xn, epi=nat.EPI_STORE_F32, yf=self.logits, x_packed=pa)
        nbytes += self.V * D * es + B * D * es + B * self.V * 4
        return n + 1, nbytes

    def decode(self, max_new: int, sp: dict, force_stop=None, use_graph=True, check_every=16, return_logits=False):
        """Run the sampling loop after prefill().  Returns codes int64 [B, n] padded with the stop token
        (HF generate semantics: rows that emitted EOS keep emitting pad = EOS)."""
        B = self._B
        if self._shared_prefix is not None:
            raise ValueError("decode(): prefill(beams=n) cached the prompt once per batch element; only decode_beam() can follow it")
        if self.kv is None and self._S + max_new + 1 > self._cap_s:
            raise ValueError("decode(): max_new exceeds the capacity reserved by prefill()")
        if self.kv is not None:
            for b in range(B):                                   # (no-op when prefill() was given the same max_new)
                self.kv.cover(b, self._pad_host[b], self._S + max_new + 1)
            self.kv.flush()
        if force_stop is None:
            self.force_stop[:B] = -1
        else:
            self.force_stop[:B] = torch.as_tensor(force_stop, dtype=torch.int32).to(self.device)
        logits_trace = [self.logits[:B].clone()] if return_logits else None
        sp = self._seed_to_state(sp)
        self._sample(B, sp)  # token 1 from the prefill logits
        n = 1
        G = 1 if return_logits else self.steps_per_graph
        while n < max_new:
            if use_graph and not self.force_eager and n >= 2:
                k = G if n + G <= max_new else 1          # several tokens per replay while they fit
                self._get_graph(B, sp, k).replay()
            else:
                k = 1
                self._step_kernels(B, sp)                 # eager: first step doubles as the warm-up before capture
            prev = n
            n += k
            if return_logits:
                logits_trace.append(self.logits[:B].clone())
            if n // check_every != prev // check_every and self._poll() >= B:
                break
        self._poll()
        codes = self.history[:B, :n].to(torch.int64)
        return (codes, torch.stack(logits_trace, 0)) if return_logits else codes

    # ------------------------------------------------------------------------------------------------ slot refill
    def _stage(self, rows, prefixes, stops, n):
        """Prepare new utterances for the decode slots `rows`, to JOIN a running loop at the moment every row has been given
        n tokens' worth of steps (state[0] = n - 1 with the bump pending, the next step writes cache position S+n-1).
        prefixes: fp32 [P_j, D] each (cond | text, no padding).  A new row is laid out exactly like an initial row of a batch
        whose loop started n - 1 steps later: prompt + start token at cache positions [pad, S+n-2] of its slot, left padding
        pad = S+n-1-(P_j+1).  One packed prefill pass over all new rows; their keys / values are kept aside ([M, L, H, 64]) and
        the logits of their last positions computed on buffers of their own -- nothing the running loop reads or writes is
        touched, so this may run on another stream beside the loop's steps."""
        import numpy as np
        T, D, H, dev = self.dtype, self.D, self.H, self.device
        k = len(rows)
        end = self._S + n - 1                              # one past the last prompt position
        lens = [int(p.shape[0]) + 1 for p in prefixes]
        if max(lens) > end:
            raise ValueError("decode_refill(): a new prompt is longer than the positions the loop has passed")
        pads = [end - L for L in lens]
        start = (self.mel_emb[self.start_mel] + self.mel_pos[0])[None]
        h = torch.cat([t for p in prefixes for t in (p.to(dev, torch.float32), start)], dim=0).contiguous()
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        M = int(off[-1])
        slot = np.concatenate([np.full(L, r) for L, r in zip(lens, rows)])
        posi = np.concatenate([pd + np.arange(L) for L, pd in zip(lens, pads)])
        if self.kv is not None:                            # paged cache: (block, offset) of every staged position (the caller
            slot, posi = self.kv.phys(slot, posi)          # has dealt the new rows their blocks already)
        meta = torch.from_numpy(np.concatenate([slot, posi, off, off[1:] - 1, np.asarray(rows), np.asarray(pads),
                                                np.asarray(stops)]).astype(np.int64)).to(dev)     # one upload
        st = {"k": k, "n": n, "i_b": meta[:M], "i_p": meta[M:2 * M],
              "i_rows": meta[2 * M + 2 * k + 1:2 * M + 3 * k + 1],
              "pads": meta[2 * M + 3 * k + 1:2 * M + 4 * k + 1].to(torch.int32), "stops": meta[2 * M + 4 * k + 1:].to(torch.int32)}
        row_off = meta[2 * M:2 * M + k + 1].to(torch.int32)
        last = meta[2 * M + k + 1:2 * M + 2 * k + 1]
        kst = torch.empty(M, self.L, H, 64, dtype=T, device=dev)
        vst = torch.empty(M, self.L, H, 64, dtype=T, device=dev)
        Smax = max(lens)

        def attn(i, qkv, att):     # keys / values are kept aside (they enter the cache when the rows join), plain causal attention
            q4 = qkv.view(M, 3, H, 64)
            kst[:, i] = q4[:, 1]
            vst[:, i] = q4[:, 2]
            nat.attn_prefill_packed(qkv, att, None, None, row_off, None, k, Smax, H, self._cap_s)
        h = self._big_m_layers(h, attn)
        xn_t = torch.zeros(nat.packed_rows(k), D, dtype=T, device=dev)
        lg_t = torch.empty(k, self.V, dtype=torch.float32, device=dev)
        nat.ln_reduce(h[last].contiguous(), self.ln_f[0], self.ln_f[1], xn_t, w2=self.final_norm[0], b2=self.final_norm[1],
                      y_packed=self.pa)
        nat.gemm_skinny(T, k, self.V, D, self.w_head, self.b_head, x=xn_t, epi=nat.EPI_STORE_F32, yf=lg_t, x_packed=self.pa)
        st.update(kst=kst, vst=vst, logits=lg_t)
        return st

    def _join(self, st, sp):
        """The staged rows enter the loop (between two of its steps, at the n they were staged for): keys / values into the
        slots' cache rows, the first token of every new row sampled from its prefill logits -- on buffers of its own, the
        running rows' logits stay --, and the per-row state: left padding, stop step, own clock row_step0 = n - 1 (mel
        positions, history index and the repetition-penalty window count from the row's own first token)."""
        k, n, dev = st["k"], st["n"], self.device
        self.kc[:, st["i_b"], :, st["i_p"]] = st["kst"]    # [M, L, H, 64] -> positions [pad, S+n-2] of the slots (rows | blocks)
        self.vc[:, st["i_b"], :, st["i_p"]] = st["vst"]
        tok_t = torch.zeros(k, dtype=torch.int32, device=dev)
        hist_t = torch.zeros(k, 8, dtype=torch.int32, device=dev)
        fin_t = torch.zeros(k, dtype=torch.int32, device=dev)
        step0_t = torch.full((k,), n - 1, dtype=torch.int32, device=dev)
        self.state[2:3] -= k                               # these slots were counted as finished
        nat.sample(st["logits"], tok_t, hist_t, fin_t, self.state, self.extra_ids, st["stops"], sp["repetition_penalty"],
                   sp["temperature"], sp["top_k"], sp["top_p"], sp["do_sample"], sp["seed"], self.stop_mel, None, no_advance=True,
                   row_step0=step0_t)
        i_rows = st["i_rows"]
        self.tokens[i_rows] = tok_t
        self.history[i_rows, 0] = tok_t
        self.finished[i_rows] = fin_t
        self.force_stop[i_rows] = st["stops"]
        self.row_step0[i_rows] = step0_t
        self.pad[i_rows] = st["pads"]

    def decode_refill(self, max_new: int, sp: dict, feed, force_stop=None, use_graph=True, check_every=16, positions=None,
                      staged=True):
        """Continuous batching: the sampling loop after prefill(), with every slot whose row has emitted its stop token
        refilled from a queue (the mitigation for mixed output lengths).  num_beams = 1 only.
        feed(k) -> up to k items (prefix_emb fp32 [P, D] = cond | text without padding, stop step or -1); fewer than k means
        the queue is empty.  Rows are numbered in the order they entered: 0..B-1 = the prefilled batch, then the fed ones.
        max_new bounds every row's own length (a row that reaches it is stopped there).  Returns (codes, leftover): codes[id]
        int64 [n_id] ends with the stop token; leftover = items that were fed but could not be placed any more because the
        cache positions reserved by prefill() -- or the smaller budget `positions` -- ran out (the caller starts a new loop
        with them).
        Every check_every steps the host reads the `finished` flags (the loop's one synchronisation).  staged = True: the
        prompts of the utterances that take the freed slots are prefilled on a SECOND STREAM while the loop runs its next
        check_every steps, and join at the following poll -- the ~300 launches of that pass are enqueued and executed under
        the loop's own steps instead of stalling it (the slot idles check_every steps longer).  staged = False: prefill and
        join at once, the loop waits.
        A row's tokens are those it would get decoded alone with the same logits (greedy: identical codes up to the usual
        reduction-order noise of a different left padding); sampled rows draw from the loop's Philox stream (row slot, loop
        step), so they differ from a stand-alone run as two seeds do."""
        B, S = self._B, self._S
        if self._shared_prefix is not None or self._kv_rows is not None:
            raise ValueError("decode_refill(): num_beams = 1 only")
        kv = self.kv
        if kv is not None:
            # paged cache: the loop's position counter may grow for as long as the queue lasts -- a row only has to keep its own
            # window (prompt + max_new + the steps up to the next poll) inside the block table's ring
            limit = (1 << 30) if positions is None else int(positions)
            if max_new + 2 * check_every + 2 > kv.window:
                raise ValueError("decode_refill(): max_new does not fit the block table's window")
        else:
            limit = self._cap_s if positions is None else min(self._cap_s, int(positions))
        if S + max_new + check_every > limit:
            # the loop runs whole blocks of check_every steps before it looks at the flags again: the last rows can take
            # the loop check_every - 1 steps past max_new, and every step appends one K / V position for every slot
            raise ValueError("decode_refill(): the position budget must hold max_new + check_every positions behind the prompt "
                             "(prefill(max_new + check_every))")
        dev = self.device
        fs = [-1] * B if force_stop is None else [int(v) for v in force_stop]
        fs = [max_new - 1 if v < 0 else min(v, max_new - 1) for v in fs]
        self.force_stop[:B] = torch.tensor(fs, dtype=torch.int32).to(dev)
        sp = self._seed_to_state(sp)
        self.kv_share.zero_()                                # a refilled row 0 no longer holds the shared block where the others expect it
        owner, start = list(range(B)), [0] * B              # owner: utterance id | None (free) | -1 (reserved for staged rows)
        next_id, codes, leftover, fed_out = B, {}, [], False
        stats = self.refill_stats = {"steps": 0, "polls": 0, "refill_calls": 0, "rows_refilled": 0, "staged": bool(staged)}
        main = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev) if staged else None
        pending = None                                      # (staged rows, slots, event)
        stop_id = self.stop_mel
        pads = list(self._pad_host[:B])                      # left padding per slot (paged cache: where a row's window starts)
        if kv is not None:
            stats["blocks"], stats["peak_blocks"] = kv.blocks - 1, kv.used_blocks()
        self._sample(B, sp)
        n = 1
        while True:
            # ---- A: rows staged during the last steps join here
            if pending is not None:
                st, rows, ev = pending
                main.wait_event(ev)
                self._join(st, sp)
                for t in st.values():                       # allocated on the side stream, last used on this one
                    if torch.is_tensor(t):
                        t.record_stream(main)
                for r in rows:
                    owner[r], start[r] = next_id, n - 1
                    next_id += 1
                pending = None
            # ---- B: utterances for the slots that are free now
            free = [r for r in range(B) if owner[r] is None]
            items, n_join = [], n + check_every if staged else n
            if free and not fed_out and n > 1:
                items = list(feed(len(free)))
                if len(items) > len(free):
                    raise ValueError(f"decode_refill(): feed({len(free)}) returned {len(items)} items")
                if len(items) < len(free):
                    fed_out = True
                if items and limit - (S + n_join + 1) < max_new + check_every:    # steps the loop could still take
                    leftover, items, fed_out = items, [], True
                if items and kv is not None:
                    # blocks for the new rows' windows: prompt + start token in front of the join position, then max_new tokens
                    # and the steps up to the poll after the row's last one
                    end = S + n_join - 1
                    need = sum((end + max_new + 2 * check_every) // kv.bs - (end - int(p.shape[0]) - 1) // kv.bs + 1 for p, _ in items)
                    if need > len(kv.free):
                        leftover, items, fed_out = items, [], True          # (a pool sized by prefill(slots_window=...) never gets here)
            if items:
                rows = free[: len(items)]
                stops = [max_new - 1 if int(v) < 0 else min(int(v), max_new - 1) for _, v in items]
                if kv is not None:
                    end = S + n_join - 1
                    for r, (p, _) in zip(rows, items):
                        pads[r] = end - (int(p.shape[0]) + 1)
                        kv.cover(r, pads[r], end + check_every + 1)
                stats["refill_calls"] += 1
                stats["rows_refilled"] += len(rows)
                if not staged:
                    self._join(self._stage(rows, [p for p, _ in items], stops, n), sp)
                    for r in rows:
                        owner[r], start[r] = next_id, n - 1
                        next_id += 1
                    items = []
                else:
                    fed = torch.cuda.Event()
                    fed.record(main)                        # the prefix embeddings are complete on the loop's stream
            if fed_out and pending is None and not items and all(o is None for o in owner):
                break
            # ---- C: check_every steps of the loop (paged cache: every live row's blocks reach past the last of them)
            if kv is not None:
                for r in range(B):
                    if owner[r] is not None and kv.span[r] is not None:
                        kv.cover(r, pads[r], S + n - 1 + check_every + 1)
                kv.flush()
                stats["peak_blocks"] = max(stats["peak_blocks"], kv.used_blocks())
            todo = check_every
            while todo > 0:
                G = self.steps_per_graph
                if use_graph and not self.force_eager and n >= 2:
                    kk = G if G <= todo else 1
                    self._get_graph(B, sp, kk).replay()
                else:
                    kk = 1
                    self._step_kernels(B, sp)
                n += kk
                todo -= kk
            # ---- D: the new rows' prompts, enqueued behind the steps on the host and run beside them on the GPU
            if items:
                with torch.cuda.stream(side):
                    side.wait_event(fed)
                    for p, _ in items:
                        p.record_stream(side)               # allocated on the loop's stream, read on this one
                    st = self._stage(rows, [p for p, _ in items], stops, n_join)
                    ev = torch.cuda.Event()
                    ev.record(side)
                for r in rows:
                    owner[r] = -1
                pending = (st, rows, ev)
            # ---- E: one host synchronisation: which rows have stopped
            stats["steps"], stats["polls"] = n, stats["polls"] + 1
            self._poll()
            fin = self.finished[:B].tolist()
            newly = [r for r in range(B) if fin[r] and owner[r] is not None and owner[r] >= 0]
            if newly:
                width = max(n - start[r] for r in newly)
                hist = self.history[torch.tensor(newly, device=dev), :width].cpu()
                for j, r in enumerate(newly):
                    row = hist[j, : n - start[r]].to(torch.int64)
                    hit = (row == stop_id).nonzero()
                    codes[owner[r]] = row[: int(hit[0]) + 1] if hit.numel() else row
                    owner[r] = None
                    if kv is not None:
                        kv.release(r)      # its blocks go back to the pool; the slot's later (formal) appends land in the scratch block
        return [codes[i] for i in range(next_id)], leftover

    # ------------------------------------------------------------------------------------------------ beam search
    def _ensure_beam(self, B: int, nb: int):
        R, dev = B * nb, self.device
        if getattr(self, "_beam_cap", (0, 0, 0)) == (B, nb, self._cap_s):
            return
        # the buffers below are about to be replaced: every captured beam step holds their addresses, so those graphs
        # must go with them (a B=4 -> B=3 -> B=4 sequence would otherwise replay a graph over freed memory)
        for key in [k for k in self._graphs if k and k[0] == "beam"]:
            del self._graphs[key]
        cap = self.history.shape[1]
        self.b_scores = torch.zeros(R, dtype=torch.float32, device=dev)
        self.b_src = torch.zeros(R, dtype=torch.int32, device=dev)
        self.b_hist = torch.zeros(2, R, cap, dtype=torch.int32, device=dev)
        self.b_hyp_score = torch.zeros(B, nb, dtype=torch.float32, device=dev)
        self.b_hyp_len = torch.zeros(B, nb, dtype=torch.int32, device=dev)
        self.b_hyp_tok = torch.zeros(B, nb, cap, dtype=torch.int32, device=dev)
        self.b_n_hyp = torch.zeros(B, dtype=torch.int32, device=dev)
        self.b_worst = torch.zeros(B, dtype=torch.float32, device=dev)
        self.b_done = torch.zeros(B, dtype=torch.int32, device=dev)
        self.b_kv_rows = torch.zeros(2, R, self._cap_s, dtype=torch.int32, device=dev)   # [parity][logical row][position]
        self.b_scratch = nat.beam_scratch(R, dev)   # per-row candidates handed from the step's first launch to its second
        self._beam_cap = (B, nb, self._cap_s)

    def _beam_select(self, B, nb, sp):
        R = B * nb
        self._pending_bump = False   # the beam step kernel advances step counter and cache position itself
        nat.beam_step(self.logits[:R], nb, self.tokens, self.b_src, self.b_scores, self.b_hist, self.b_hyp_score, self.b_hyp_len,
                      self.b_hyp_tok, self.b_n_hyp, self.b_worst, self.b_done, self.state, self.extra_ids,
                      sp["repetition_penalty"], sp["temperature"], sp["top_k"], sp["top_p"], sp["do_sample"],
                      sp.get("length_penalty", 0.0), sp["seed"], self.stop_mel, scratch=self.b_scratch)
        if self._kv_rows is not None:
            nat.beam_kv_rows(self._kv_rows, self.b_src, self.state)
        else:
            nat.beam_reorder_kv(self.kc, self.vc, self.b_src, self.state, B, nb)

    def _step_kernels_beam(self, B, nb, sp):
        self._step_transformer(B * nb)
        self._beam_select(B, nb, sp)

    def decode_beam(self, max_new: int, sp: dict, num_beams: int, use_graph=True, check_every=16, num_return_sequences=1):
        """Beam search / beam-sample after prefill() of B*num_beams rows (row = b*num_beams + beam, the beams of a batch
        element start as copies).  HF 4.44.2 semantics (oracle/beam_ref.py); returns int64 [B * num_return_sequences, n]: the
        num_return_sequences best hypotheses of every element, best first (row = b * num_return_sequences + rank),
        right-padded with the stop token."""
        nb = int(num_beams)
        if not 1 <= int(num_return_sequences) <= nb:
            raise ValueError("num_return_sequences has to be in [1, num_beams]")   # generate() raises the same
        R = self._B
        if self.kv is not None:
            raise ValueError("decode_beam(): beam search addresses whole contiguous cache rows: prefill(..., paged=False) or prefill(beams=n)")
        if self._shared_prefix is not None and self._shared_prefix[1] != nb:
            raise ValueError("decode_beam(): prefill(beams=...) was given another beam count")
        assert R % nb == 0, "prefill() must have been given B*num_beams rows"
        B = R // nb
        if self._S + max_new + 1 > self._cap_s:
            raise ValueError("decode_beam(): max_new exceeds the capacity reserved by prefill()")
        self._ensure_beam(B, nb)
        self.kv_share.zero_()       # beam reorders (copy mode) move cache rows: the promise behind the shared reads does not hold
        self.b_scores.zero_()
        self.b_scores.view(B, nb)[:, 1:] = -1e9
        self.b_hist.zero_()
        self.b_n_hyp.zero_()
        self.b_worst.fill_(1e9)
        self.b_done.zero_()
        if self._shared_prefix is not None and self._shared_prefix != (B, nb):
            raise ValueError("decode_beam(): prefill(beams=...) was given another batch / beam count")
        if self.beam_kv == "table":
            rows = torch.arange(R, dtype=torch.int32, device=self.device)
            self.b_kv_rows[0] = rows[:, None]           # identity: every row holds itself ...
            if self._shared_prefix is not None:         # ... except the prompt, cached once per batch element in row b
                self.b_kv_rows[0][:, : self._S] = (rows // nb)[:, None]
            self._kv_rows = self.b_kv_rows
        else:
            self._kv_rows = None
        try:
            return self._decode_beam_loop(B, nb, max_new, sp, use_graph, check_every, int(num_return_sequences))
        finally:
            self._kv_rows = None

    def _decode_beam_loop(self, B, nb, max_new, sp, use_graph, check_every, num_return=1):
        sp = self._seed_to_state(sp)
        self._beam_select(B, nb, sp)  # token 1 from the prefill logits
        n = 1
        key = ("beam", B, nb, self.beam_kv, self.decode_mode, self.lora, tuple(self.fold_rows), self.fold_rows_consumers, self.fold_wide,
               self.pa, self.KSPLIT,
               tuple(sorted(sp.items())))
        while n < max_new:
            if use_graph and not self.force_eager and n >= 2:
                g = self._graphs.get(key)
                if g is None:
                    g = torch.cuda.CUDAGraph()
                    with nat.CAPTURE_LOCK, torch.cuda.graph(g):
                        self._step_kernels_beam(B, nb, sp)
                    self._graphs[key] = g
                g.replay()
            else:
                self._step_kernels_beam(B, nb, sp)
            n += 1
            if n % check_every == 0 and self._poll() >= B:
                break
        self._poll()
        return self._beam_finalize(B, nb, n, float(sp.get("length_penalty", 0.0)), max_new, num_return)

    def _beam_finalize(self, B, nb, n, length_penalty, max_new, num_return=1):
        """BeamSearchScorer.finalize on the host: running beams of unfinished batch elements become hypotheses (score =
        sum_logprobs / generated_len**length_penalty), the num_return best hypotheses of each element are returned, best
        first (num_beam_hyps_to_keep = num_return_sequences; the device store already holds num_beams per element)."""
        hs = self.b_hyp_score.cpu().numpy()
        hl = self.b_hyp_len.cpu().numpy()
        ht = self.b_hyp_tok.cpu().numpy()
        nh = self.b_n_hyp.cpu().numpy()
        done = self.b_done.cpu().numpy()
        sc = self.b_scores.cpu().numpy().reshape(B, nb)
        hist = self.b_hist[n & 1].cpu().numpy().reshape(B, nb, -1)
        best = []
        for b in range(B):
            hyps = [(float(hs[b, i]), i, ht[b, i, : hl[b, i]].tolist()) for i in range(int(nh[b]))]
            if not done[b]:
                worst = min((h[0] for h in hyps), default=1e9)
                for k in range(nb):
                    score = float(sc[b, k]) / (float(n) ** length_penalty if length_penalty != 0.0 else 1.0)
                    if len(hyps) < nb or score > worst:
                        hyps.append((score, nb + k, hist[b, k, :n].tolist()))
                        if len(hyps) > nb:
                            hyps.remove(min(hyps, key=lambda h: (h[0], h[1])))
                        worst = min(h[0] for h in hyps)
            ranked = sorted(hyps, key=lambda h: (h[0], h[1]), reverse=True)
            best.extend(h[2] for h in ranked[:num_return])
        width = min(max(len(t) for t in best) + 1, max_new)  # sent_max_len = min(longest + 1, max_length)
        out = torch.full((len(best), width), self.stop_mel, dtype=torch.int64)
        for b, t in enumerate(best):
            out[b, : min(len(t), width)] = torch.tensor(t[:width], dtype=torch.int64)
        return out.to(self.device)

    def _seed_to_state(self, sp):
        """The draw key is launch-argument seed + state[4..5]; the loop keeps the argument at 0 and the real seed in the
        device state, so one captured step serves every seed (no re-capture per call)."""
        seed = int(sp.get("seed", 0)) & 0xFFFFFFFFFFFFFFFF
        lo, hi = seed & 0xFFFFFFFF, seed >> 32
        as_i32 = lambda v: v - (1 << 32) if v >= (1 << 31) else v  # noqa: E731
        self.state[4:6] = torch.tensor([as_i32(lo), as_i32(hi)], dtype=torch.int32).to(self.device)
        out = dict(sp)
        out["seed"] = 0
        return out

    def _get_graph(self, B, sp, nsteps=1):
        # everything the captured launches depend on besides the buffers: a knob toggled after a capture must not replay the
        # old variant (skip_finished: whether the attention is given the finished flags)
        key = (B, nsteps, self.skip_finished, self.decode_mode, self.lora, tuple(self.fold_rows), self.fold_wide, self.pa, self.KSPLIT,
               self.share_kv_reads, None if self.kv is None else self.kv.bs, tuple(sorted(sp.items())))
        g = self._graphs.get(key)
        if g is None:
            g = torch.cuda.CUDAGraph()
            with nat.CAPTURE_LOCK, torch.cuda.graph(g):  # no other thread may allocate / synchronise during a capture
                for _ in range(nsteps):
                    self._step_kernels(B, sp)
            self._graphs[key] = g
        return g

    def step_bytes(self, B: int, ctx: int) -> int:
        """Algorithmic HBM bytes of one decode step: weights once + KV read + KV append."""
        es = 4 if self.dtype == torch.float32 else 2
        kv = self.L * 2 * self.D * es
        return self.weight_bytes + B * ctx * kv + B * kv

"""Multi-GPU execution: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference is single-GPU (run.sh:29); what shards here is the structure of the hot path itself
(SURVEY.md 8e):
  * frames / clips are independent  -> `shard_slice` + `predict_sharded`: pure data parallelism, no collective
    on the data path (used by bench.py and the pose-only evaluations);
  * inside one clip every layer before the action head is frame independent (TimeDistributed,
    layers.py:63-104; inference BatchNorm) -> `split_frames` cuts a clip model into a FRAME stage (runs on
    T/G frames per rank) and a HEAD stage (needs all T frames), and `ShardedClipModel` joins them with ONE
    all-gather of a packed [N, T/G, J, C_packed] fp32 buffer per clip batch.  Payloads are <= ~1 MB per clip,
    so the exchange is latency bound; ring vs direct does not matter at this size.
"""
import numpy as np

from . import graph as G
from . import layers as L
from .model import Model

# number of trailing dims an op actually couples; everything in front of them is batch for that op
_CORE_RANK = {
    'conv': 3, 'sepconv': 3, 'pool': 3, 'upsample': 3, 'zeropad': 3, 'softmax2d': 3, 'expect2d': 3,
    'jointprob': 3, 'kronecker': 3, 'depthmean': 3, 'depthsum': 3, 'globalmax2d': 3, 'globalmaxmin': 3,
    'context_agg': 2, 'softargmax1d': 2, 'globalmax1d': 2,
    'relu': 1, 'bn': 1, 'add': 1, 'mul': 1, 'sigmoid': 1, 'scale': 1, 'concat': 1, 'slice': 1, 'softmax': 1,
}


def shard_slice(n, rank, world):
    """Contiguous [lo, hi) share of n items for `rank` (the first n % world ranks get one extra)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def predict_sharded(model, x, rank, world, batch_size=32):
    """Data-parallel predict: this rank runs its contiguous share of the batch, nothing is exchanged."""
    lo, hi = shard_slice(len(x), rank, world)
    return model.predict(x[lo:hi], batch_size=batch_size), (lo, hi)


# ---------------------------------------------------------------------------------------------------------
# graph partition
# ---------------------------------------------------------------------------------------------------------

def frame_level_nodes(model):
    """uids of nodes that treat the clip's frame axis (leading dim of the single model input) purely as batch."""
    t_in = model.inputs[0]
    frame_t = {t_in.uid}
    frame_n = set()
    for n in model._nodes:
        core = _CORE_RANK.get(n.op)
        if n.op == 'reshape':
            ok = len(n.inputs[0].shape) >= 2 and len(n.outputs[0].shape) >= 2 and \
                n.inputs[0].shape[0] == n.outputs[0].shape[0]
        else:
            ok = core is not None and all(len(t.shape) > core for t in n.inputs)
        if ok and all(t.uid in frame_t for t in n.inputs):
            frame_n.add(n.uid)
            for o in n.outputs:
                frame_t.add(o.uid)
    return frame_n, frame_t


def split_frames(model, shards):
    """Cut a clip model (single input [T, H, W, C]) into (frame_model, head_model, info).

    frame_model : input [T/shards, H, W, C] -> ONE packed output [T/shards, J, C_packed] holding, side by side
                  on the channel axis, every tensor the head stage or the caller needs from the frame stage
    head_model  : inputs = those tensors at full T (or None when the model has no temporal part)
    info        : dict(cut=[(shape_without_T, channel_offset, channels)], passthrough={output index: cut index},
                       head_outputs=[output indices produced by the head])
    """
    if len(model.inputs) != 1:
        raise ValueError('split_frames expects a single-input clip model')
    T = model.inputs[0].shape[0]
    if T % shards:
        raise ValueError('clip length %d does not divide into %d shards' % (T, shards))
    frame_n, frame_t = frame_level_nodes(model)
    consumers = {}
    for n in model._nodes:
        for t in n.inputs:
            consumers.setdefault(t.uid, []).append(n)
    cut, seen = [], set()
    for n in model._nodes:                        # deterministic order: graph order
        if n.uid not in frame_n:
            continue
        for o in n.outputs:
            used_by_head = any(c.uid not in frame_n for c in consumers.get(o.uid, []))
            is_out = any(o.uid == t.uid for t in model.outputs)
            if (used_by_head or is_out) and o.uid not in seen:
                seen.add(o.uid)
                cut.append(o)
    if not cut:
        raise ValueError('model has no frame-level stage')
    lead = cut[0].shape[:-1]
    if any(t.shape[:-1] != lead for t in cut):
        raise NotImplementedError('cut tensors with different pixel shapes: %s' % [t.shape for t in cut])

    # frame stage, re-built on T/shards frames (shapes: replace the leading T)
    Tl = T // shards
    x_local = L.Input((Tl,) + model.inputs[0].shape[1:])
    local = G.clone_subgraph(model.inputs, cut, [x_local], relead=(T, Tl))
    packed = L.concatenate(local) if len(local) > 1 else local[0]
    frame_model = Model(x_local, packed, name=(model.name or 'model') + '_frames')

    offs, off = [], 0
    for t in cut:
        offs.append((t.shape[1:], off, t.shape[-1]))
        off += t.shape[-1]
    passthrough, head_outputs = {}, []
    cut_index = {t.uid: i for i, t in enumerate(cut)}
    for k, t in enumerate(model.outputs):
        if t.uid in cut_index:
            passthrough[k] = cut_index[t.uid]
        else:
            head_outputs.append(k)
    head_model = None
    if head_outputs:
        new_in = [L.Input(t.shape) for t in cut]
        outs = G.clone_subgraph(cut, [model.outputs[k] for k in head_outputs], new_in, allow_unused=True)
        head_model = Model(new_in, outs, name=(model.name or 'model') + '_head')
    info = dict(cut=offs, passthrough=passthrough, head_outputs=head_outputs, packed_channels=off, T=T, Tl=Tl)
    return frame_model, head_model, info


# ---------------------------------------------------------------------------------------------------------
# runtime
# ---------------------------------------------------------------------------------------------------------

def all_gather_rank_major(local, group=None, world=None, out=None, always=False):
    """ONE collective: all-gather a [N, T_local, ...] tensor -> [G, N, T_local, ...] (rank-major, exactly what
    `all_gather_into_tensor` writes; no re-ordering pass).  Works on CUDA tensors (RCCL) and CPU tensors (gloo).
    `out`: optional persistent destination of that shape.  With one rank the result is a view of `local` -- unless
    `always`: then the collective is issued even for a world of one (the RCCL call path, the persistent buffer and the
    stream ordering around it are what a one-GPU test box can exercise of the multi-GPU leg)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if world is None else world
    if world == 1 and not always:
        return local.unsqueeze(0)
    local = local.contiguous()
    shape = (world,) + tuple(local.shape)
    if out is None or tuple(out.shape) != shape or out.dtype != local.dtype or out.device != local.device:
        out = torch.empty(shape, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.view((world * local.shape[0],) + tuple(local.shape[1:])), local, group=group)
    return out


def frames_view(gathered):
    """[G, N, Tl, ...] -> the strided VIEW [N, G, Tl, ...]; flattening (G, Tl) gives the clip's frame axis in order.
    A consumer that copies it into its own contiguous [N, T, ...] input (Executor.run_device) does the re-ordering
    inside a copy it makes anyway."""
    return gathered.permute((1, 0) + tuple(range(2, gathered.dim())))


def all_gather_frames(local, group=None, world=None):
    """All-gather a [N, T_local, ...] tensor along the frame axis -> contiguous [N, T, ...] (one collective + one
    re-ordering copy; the device-resident runtime below avoids the copy, this form serves host-side callers)."""
    g = all_gather_rank_major(local, group, world)
    n, tl = local.shape[0], local.shape[1]
    return frames_view(g).reshape((n, g.shape[0] * tl) + tuple(local.shape[2:]))


class ShardedClipModel:
    """Frame-sharded execution of a clip model over the ranks of a process group.

    Default (HIP) path, device resident end to end: this rank's frames [N, T/G, H, W, C] (host array or device tensor)
    -> frame stage (hipGraph replay, `Executor.run_device`) -> packed device view [N, T/G, J, Cp] -> ONE all-gather
    (RCCL over xGMI) into a persistent [G, N, T/G, J, Cp] buffer -> the head stage's inputs are strided channel
    slices of that buffer, re-ordered to [N, T, J, c] by the device-to-device copy into the head's plan (no separate
    re-ordering pass) -> head stage (hipGraph replay) -> device views of the outputs; only `predict` copies results
    to the host.

    Streams, serial form (`overlap=False`): the frame stage runs on its executor's stream F, the collective on torch's
    current stream C, the head stage on its executor's stream H.  Per step: F waits for C (the caller's input, and the previous
    collective's read of the frame arena) and for the event E recorded after H's input copies of the previous step (with one
    rank H reads the frame arena directly); C waits for F and for E (the gather buffer is re-used); H waits for C -- the
    frame stage of step i + 1 cannot start before the collective of step i has read the frame arena: the transfer is
    exposed.

    [r06] Pipelined form (`overlap=True`, the default of the HIP stages): the packed tensor of step i is copied (device to
    device, on F) into one of TWO send buffers, the collective runs on its own stream K from that buffer into one of two
    gather buffers, the head stage reads the gather buffer on H.  Slot k = i mod 2: F's copy into send[k] waits for the
    collective and the head that last used the slot, K waits for F's copy and for the head's input copies of step i - 2,
    H waits for K.  F never waits for K or H of the step before, so the frame stage of step i + 1 (convolutions: 99 % of
    a step) runs WHILE the all-gather of step i crosses xGMI and its head stage runs: transfer and head hide behind
    conv work whenever steps are issued back to back (bench.py, a serving loop).  Results are bit-identical to the serial
    form; `last_timing` holds the events a caller can read the exposed time from.

    frame_fn(x_local [N, T/G, H, W, C]) -> packed [N, T/G, J, Cp] (torch tensor, any device)
    head_fn(list of [N, T, J, c_i])     -> list of arrays / tensors
    may be injected: the CPU tests put oracle stand-ins there to exercise the collective and the bookkeeping with the
    gloo backend."""

    def __init__(self, model, rank=None, world=None, group=None, frame_fn=None, head_fn=None, always_collective=False,
                 overlap=None):
        self.always_collective = bool(always_collective)     # issue the all-gather even for a world of one rank
        # pipelined streams (class docstring): on for the HIP stages unless asked otherwise; injected CPU stages run serially
        self.overlap = (frame_fn is None and head_fn is None) if overlap is None else bool(overlap)
        if self.overlap and (frame_fn is not None or head_fn is not None):
            raise ValueError('overlap=True needs the HIP stages (it orders HIP streams)')
        if rank is None or world is None:
            import torch.distributed as dist
            rank = dist.get_rank(group) if rank is None else rank
            world = dist.get_world_size(group) if world is None else world
        self.group = group
        self.rank, self.world = rank, world
        self.model = model
        self.frame_model, self.head_model, self.info = split_frames(model, self.world)
        for stage in (self.frame_model, self.head_model):          # the stages run with the clip model's engine options
            if stage is not None:
                stage.num_streams, stage.gemm_precision = model.num_streams, model.gemm_precision
                stage.stream_policy = model.stream_policy
        if self.head_model is not None:
            self.head_model._stream_role = 'head'                  # (its executor is created on first use: engine/executor.py shared_stream)
        self.frame_fn = frame_fn or self._frame_hip
        self.head_fn = head_fn or self._head_hip
        self.last_outputs = None
        self._gather_buf = None
        self._head_inputs_read = None        # event: the head stage's input copies of the previous step are done
        # pipelined form: two slots of {send buffer, gather buffer, events}, a stream for the collective, a step counter
        self._slots = [dict(send=None, gather=None, sent=None, gathered=None, read=None) for _ in range(2)]
        self._comm_stream = None
        self._step = 0
        self.last_timing = None              # (frame stage done, collective start, collective end) events of the last step

    # -- default stages on the GPU -----------------------------------------------------------------------
    def _frame_hip(self, x_local):
        import torch
        ex = self.frame_model.executor
        cur = torch.cuda.current_stream()
        ex.stream.wait_stream(cur)           # x_local may come from C; C also read the frame arena (last collective)
        if self._head_inputs_read is not None:
            ex.stream.wait_event(self._head_inputs_read)
            cur.wait_event(self._head_inputs_read)          # ... and the gather buffer is about to be overwritten
        if isinstance(x_local, np.ndarray):
            x_local = torch.from_numpy(np.ascontiguousarray(x_local, dtype=np.float32)).to(ex.device)
        out = ex.run_device([x_local])[0]                       # device view of the packed tensor, on ex.stream
        cur.wait_stream(ex.stream)                              # the collective runs behind torch's current stream
        return out

    def _head_hip(self, tensors):
        import torch
        ex = self.head_model.executor
        ex.stream.wait_stream(torch.cuda.current_stream())      # ... and the head stage behind the collective
        if self._head_inputs_read is None:
            self._head_inputs_read = torch.cuda.Event()
        return ex.run_device(tensors, inputs_copied=self._head_inputs_read)

    def _forward_pipelined(self, x_local, events=None):
        """[r06] One clip batch on the pipelined streams (class docstring): nothing here makes the frame stage of the NEXT
        call wait for this call's collective or head stage."""
        import torch
        info = self.info
        fx = self.frame_model.executor
        F = fx.stream
        cur = torch.cuda.current_stream()
        if self._comm_stream is None:
            from .engine.executor import shared_stream
            self._comm_stream = shared_stream(fx.device, 'comm')
        K = self._comm_stream
        slot = self._slots[self._step % 2]
        self._step += 1
        if isinstance(x_local, np.ndarray):
            x_local = torch.from_numpy(np.ascontiguousarray(x_local, dtype=np.float32)).to(fx.device)
        ready = torch.cuda.Event()
        ready.record(cur)                                     # the caller's input (only that: not C's older work)
        F.wait_event(ready)
        x_local.record_stream(F)                              # (allocated on the caller's stream, read on F)
        packed = fx.run_device([x_local])[0]                  # device view inside the frame arena, on F
        with torch.cuda.stream(F):
            if slot['send'] is None or tuple(slot['send'].shape) != tuple(packed.shape):
                slot['send'] = torch.empty(tuple(packed.shape), dtype=packed.dtype, device=packed.device)
                slot['sent'], slot['gathered'], slot['read'] = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
                slot['used'] = False
            if slot['used']:                                  # the collective / head stage of step i - 2 still read the slot
                F.wait_event(slot['gathered'])
                F.wait_event(slot['read'])
            slot['send'].copy_(packed, non_blocking=True)
            slot['sent'].record(F)
        K.wait_event(slot['sent'])
        if slot['used']:
            K.wait_event(slot['read'])                        # the gather buffer of the slot is about to be overwritten
        with torch.cuda.stream(K):
            if events is not None:
                events[0].record(K)
            gathered = all_gather_rank_major(slot['send'], self.group, self.world, out=slot['gather'],
                                             always=self.always_collective)                     # [G, N, Tl, J, Cp]
            if self.world > 1 or self.always_collective:
                slot['gather'] = self._gather_buf = gathered
            if events is not None:
                events[1].record(K)
            slot['gathered'].record(K)
        slot['used'] = True
        self.last_timing = (slot['sent'], slot['gathered'])
        full = frames_view(gathered)
        n, t = full.shape[0], full.shape[1] * full.shape[2]
        parts = [full[..., off:off + c] for (_, off, c) in info['cut']]
        flat = lambda v: v.reshape((n, t) + tuple(v.shape[3:]))
        outs = [None] * len(self.model.outputs)
        if self.head_model is not None:
            hx = self.head_model.executor
            hx.stream.wait_event(slot['gathered'])
            head = hx.run_device(parts, inputs_copied=slot['read'])
            for k, o in zip(info['head_outputs'], head):
                outs[k] = o
        else:
            slot['read'].record(K)
        if info['passthrough']:
            # pose tensors that cross the cut unchanged: small contiguous copies made on the head's stream (or K), so that
            # the slot can be re-used two steps later without the caller holding views into it
            side = self.head_model.executor.stream if self.head_model is not None else K
            with torch.cuda.stream(side):
                for k, ci in info['passthrough'].items():
                    outs[k] = flat(parts[ci]).contiguous()
                if self.head_model is not None:
                    slot['read'].record(side)                 # (re-recorded behind the copies: they read the slot too)
        self.last_outputs = outs
        return outs

    def synchronize(self):
        """Wait for everything forward_device enqueued (the streams it uses are its own)."""
        import torch
        for ex in (self.frame_model.executor, self.head_model.executor if self.head_model is not None else None):
            if ex is not None:
                ex.stream.synchronize()
        if self._comm_stream is not None:
            self._comm_stream.synchronize()

    def forward_device(self, x_local, events=None):
        """One clip batch, nothing leaves the device.  `events` = (start, stop) torch events recorded around the
        collective (bench.py: collective_us).  Returns the outputs in model order (pose tensors that pass straight through
        the cut are small contiguous copies, made after the head was enqueued).  The results live on the stages' own
        streams: `synchronize()` (or `predict`) before reading them from another stream."""
        hip = getattr(self.frame_fn, '__func__', None) is ShardedClipModel._frame_hip and \
            getattr(self.head_fn, '__func__', None) is ShardedClipModel._head_hip
        if self.overlap and hip:                  # (stand-in stages assigned later run the serial form)
            return self._forward_pipelined(x_local, events)
        info = self.info
        packed = self.frame_fn(x_local)
        if events is not None:
            events[0].record()
        gathered = all_gather_rank_major(packed, self.group, self.world, out=self._gather_buf,
                                         always=self.always_collective)                         # [G, N, Tl, J, Cp]
        if self.world > 1 or self.always_collective:
            self._gather_buf = gathered
        if events is not None:
            events[1].record()
        full = frames_view(gathered)                                                            # [N, G, Tl, J, Cp]
        n, t = full.shape[0], full.shape[1] * full.shape[2]
        parts = [full[..., off:off + c] for (_, off, c) in info['cut']]
        flat = lambda v: v.reshape((n, t) + tuple(v.shape[3:]))
        outs = [None] * len(self.model.outputs)
        if self.head_model is not None:
            hip_head = getattr(self.head_fn, '__func__', None) is ShardedClipModel._head_hip
            head = self.head_fn(parts if hip_head else [flat(p) for p in parts])
            for k, o in zip(info['head_outputs'], head):
                outs[k] = o
        for k, ci in info['passthrough'].items():
            outs[k] = flat(parts[ci])
        self.last_outputs = outs
        return outs

    def export_plans(self, prefix, clips, uint8=False):
        """Write the two stages as C-level plans (include/deephar_hip.h: dh_plan_create / dh_forward) for `clips` clips per
        step: '<prefix>.frames.dhplan' takes this rank's [clips, T/G, H, W, C] frames and returns ONE packed
        [clips, T/G, J, Cp] tensor; '<prefix>.head.dhplan' takes the cut tensors at full T -- channel runs of the gathered
        buffer, `info['cut']` = [(shape, channel offset, channels)] -- in that order and returns the outputs
        `info['head_outputs']`; `info['passthrough']` maps the remaining model outputs to cut tensors.  A host without Python
        runs: frame plan -> its own all-gather of the packed tensor over the ranks (rank-major) -> head plan.  Returns
        `info` (a JSON-able dict is also written to '<prefix>.cut.json')."""
        import json
        self.frame_model.export_plan(prefix + '.frames.dhplan', clips, uint8=uint8)
        if self.head_model is not None:
            self.head_model.export_plan(prefix + '.head.dhplan', clips)
        info = dict(world=self.world, T=self.info['T'], Tl=self.info['Tl'], packed_channels=self.info['packed_channels'],
                    cut=[dict(shape=list(s), offset=o, channels=c) for (s, o, c) in self.info['cut']],
                    passthrough={str(k): v for k, v in self.info['passthrough'].items()},
                    head_outputs=list(self.info['head_outputs']))
        with open(prefix + '.cut.json', 'w') as f:
            json.dump(info, f)
        return info

    def predict(self, clips):
        """clips: [N, T, H, W, C] (the same array on every rank).  Returns the model's outputs (host arrays)."""
        info = self.info
        lo = self.rank * info['Tl']
        outs = self.forward_device(np.ascontiguousarray(clips[:, lo:lo + info['Tl']]))
        if self._comm_stream is not None:
            self.synchronize()
        elif self.head_model is not None and self.head_fn == self._head_hip:
            self.head_model.executor.stream.synchronize()
        res = []
        for o in outs:
            if isinstance(o, np.ndarray):
                res.append(o)
            else:
                res.append(o.contiguous().cpu().numpy() if o.is_cuda else np.ascontiguousarray(o.numpy()))
        return res

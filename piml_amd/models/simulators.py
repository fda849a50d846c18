"""`BaseSimulator`: the reference's rollout / training driver (src/models/simulators.py:25-832)
with the same method names and arguments, whose per-step body runs on the HIP operators.

What is restated here is the CALLER side of the hot path:
lagged explicit Euler, waypoint switching, leave-scene NaN, ground-truth injection of newly
entering agents, feature recomputation, rollout / collision losses and the Adam loop.  Data
objects are duck-typed: anything exposing the attributes of the reference's `TimeIndexedPedData` /
`ChanneledTimeIndexedPedData` (ped_features, obs_features, self_features, position, velocity,
acceleration, destination, dest_idx, dest_num, waypoints, obstacles, mask_p, mask_p_pred, labels,
num_frames, time_unit, ...) works.

Differences from the reference, all behaviour-preserving for a single call:
  * no per-step host synchronisation: boolean-mask assignments (`x[mask] = y`, which run
    `nonzero()` on the host) are `torch.where`, `if torch.sum(mask) > 0` blocks are gated
    arithmetically, the NaN assertion on the predicted acceleration is checked once after the loop;
  * the four `collision_detection(...).sum(-1)` calls per training step (simulators.py:708-724)
    are two fused `ops.collision_counts` launches (no (C,N,N) matrices);
  * the caller's data tensors are never mutated -- unless `args.inplace_quirk` is set (the default
    of `piml_amd.main`): the reference keeps VIEWS into the data object as its rollout state, so every rollout leaves
    `data.dest_idx[t_start]` = the waypoint indices at the END of that rollout (simulators.py:578, 609-613, 636) and
    `data.self_features[t_start][2:-3]` = the velocity history after its first frame (:624-626, 639); the next rollout
    of the same clip / batch then STARTS from those values (agents head for their last waypoint, agents near their
    original destination leave at once).  Reproducing the reference's numbers over a multi-rollout flow (validation
    every epoch, batches reused across epochs) needs the same carry-over; it is applied after the rollout, outside
    every captured graph.
"""
import contextlib
import os
import itertools
import time
import types

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops
from ..pedestrians import Pedestrians
from . import model as MODEL

# the frames' collision counts of a training window in one launch at its end (ops.collision_counts_frames) instead of one per frame
BATCH_COUNTS = os.environ.get('PIML_BATCH_COUNTS', '1') != '0'
# inference frames of the bottleneck variants: the network's epilogue inside the integrator's launch (ops.rollout_step ksum=)
STEP_WITH_KSUM = os.environ.get('PIML_STEP_WITH_KSUM', '1') != '0'


class RolloutResult(types.SimpleNamespace):
    """What the reference returns as a `RawData` from get_multiple_rollouts (simulators.py:655-657)."""


def _gather_waypoints(waypoints, dest_idx):
    """waypoints (*c, D, N, 2), dest_idx (*c, N) -> (*c, N, 2)   (simulators.py:614-616)."""
    idx = dest_idx.unsqueeze(-2).unsqueeze(-1).expand(*dest_idx.shape[:-1], 1, dest_idx.shape[-1], 2)
    return torch.gather(waypoints, -3, idx).squeeze(-3)


class BaseSimulator(Pedestrians):
    """Same constructor contract as the reference: `args` is the argparse namespace of src/main.py."""

    # Opt-in (args.mlp_side_stream_rows = minimum number of (slice, agent) rows): run the obstacle branch of
    # the MLP on a side stream inside captured graphs.  Measured 1.3x on a 4096-agent rollout, slower on
    # the 122-agent clip.  OFF by default: two library GEMMs running concurrently can deadlock when the
    # BLAS heuristics pick stream-K style kernels (see bench.py); enable only with validated GEMM selections.
    SIDE_STREAM_MIN_ROWS = None
    # the per-frame integrator / waypoint / injection block of the fine-tuning rollout as one HIP launch each
    # way (ops.train_rollout_step); False keeps the torch-op expression of the same arithmetic
    fused_train_step = True
    fused_rollout_losses = True      # ops.rollout_losses for the mse / collision-focus sums of the training rollout (else torch operators)
    tail_in_step = os.environ.get('PIML_TAIL_IN_STEP', '1') != '0'      # the model's agent-norm tail inside the fused frame step's launches (ops.rollout_frame tail=)

    def __init__(self, args):
        super().__init__()
        self.args = args
        self._grad_sink = ops.ParamGradSink()      # the captured fine-tuning step sums its frames' weight gradients in here
        self.set_model(args)
        self.set_optimizer(args)
        self.set_scheduler(args)
        self.finetune_flag = False
        self.test_flag = False
        self.time_iter = 0.0
        self.epoch = 0
        self.batch_idx = 0
        self.collision_count = 0
        self.hard_collision_count = 0
        self._checkpoints = {}          # {finetune_flag: best state_dict of that stage}, see save_model
        n_params = int(np.sum([p.numel() for p in self.model.parameters() if p.requires_grad]))
        print('#Trainable Parameters:', n_params)

    # ---- model / optimiser selection (simulators.py:40-136) ----
    def _build(self, args, finetune):
        if args.model not in MODEL.MODEL_TABLE:
            raise NotImplementedError(f'{args.model}: only the PINNSF family is on the accelerated path')
        net = MODEL.MODEL_TABLE[args.model][1 if finetune else 0](args)
        net.fix_dest_norm = bool(getattr(args, 'fix_dest_norm', False))      # --fix_dest_norm (quirk Q2 off)
        # this class's loops read predictions[1] (the pedestrian messages) only for the L1 regulariser and the PINN-loss
        # pre-training target (:345-347, :339-341, :735-737), never predictions[2]: without those the `pinnsf` / `pinnsf_m`
        # networks may run on the agents' sums of h2 (model.messages_wanted, PIML_POOL_TRAIN: large scenes, no active dropout)
        if hasattr(net, 'messages_wanted'):
            net.messages_wanted = bool(getattr(args, 'reg_weight', 0.0) > 0 or getattr(args, 'pinnsf_interaction', 'sim') != 'sim'
                                       or getattr(args, 'messages_wanted', False))
        # ... and the auxiliary collision head's output (predictions[-1]) only for `pinnsf_bm` under collision_pred_weight > 0
        # (:348-355, :731-733; :826-830 sees all-zero records for every other model): where no loop of this class reads it the head is
        # not launched at all (model.predictions_only: forward returns None in its place -- what the inference rollouts already did)
        if hasattr(net, 'predictions_only') and not getattr(args, 'head_wanted', False) and os.environ.get('PIML_HEAD_WANTED', '0') != '1':
            net.predictions_only = not (args.model == 'pinnsf_bm' and getattr(args, 'collision_pred_weight', 0.0) > 0)
        return net.to(args.device)

    def set_model(self, args):
        self.model = self._build(args, False)

    def set_ft_model(self, args):
        self.model = self._build(args, True)

    def _const(self, dev, value):
        """A 0-dim float32 constant on `dev`, made ONCE (outside any capture: _graphed_rollout_step asks for both before it warms
        up) and never written: the zero every loss sum starts from, the one that seeds the backward pass and stands for the
        accuracy of the absent collision head -- each `torch.zeros(())` / `zero + 1.0` / autograd's own ones_like was a launch of
        the captured step (~6 us apiece at the fine-tuning loop's size: a launch-bound graph pays per node, not per byte)."""
        key = (str(dev), repr(float(value)))
        c = self._consts.get(key) if hasattr(self, '_consts') else None
        if c is None:
            if not hasattr(self, '_consts'):
                self._consts = {}
            c = self._consts[key] = torch.full((), float(value), device=dev, dtype=torch.float32)
            if float(value) == 1.0 and c.is_cuda:
                ops.register_const_one(c)       # (a loss node fed this very tensor as its upstream gradient skips its scaling launch)
        return c

    def _side_stream_ok(self, rows):
        limit = getattr(self.args, 'mlp_side_stream_rows', self.SIDE_STREAM_MIN_ROWS)
        return limit is not None and rows >= limit

    def _capturable(self):
        return str(self.args.device).startswith('cuda')        # Adam state on the device: graph-capturable

    def _adam_kw(self):
        """On the GPU: step counters on the device (capturable) and the fused multi-tensor Adam kernel -- the
        per-tensor implementation spends ~12 launch-bound kernels on each of the 26 parameter tensors, more
        than half of the kernels of a captured fine-tuning step."""
        on_gpu = self._capturable()
        fused = on_gpu and bool(getattr(self.args, 'fused_adam', 1))
        return dict(capturable=on_gpu, fused=True) if fused else dict(capturable=on_gpu)

    def _adam(self, params, **kw):
        """torch.optim.Adam -- on the GPU the same optimiser stepped by ONE launch of this package (piml_amd.optim.Adam: bitwise
        PyTorch's fused kernel, whose step is `_foreach_add_` + a multi-tensor launch, 18 us of every training step);
        PIML_ADAM=torch or args.fused_adam = 0 keep PyTorch's own step."""
        kw = dict(kw, **self._adam_kw())
        if kw.get('fused') and os.environ.get('PIML_ADAM', 'piml') != 'torch':
            from ..optim import Adam
            return Adam(params, **kw)
        return torch.optim.Adam(params, **kw)

    def set_optimizer(self, args):
        self.optimizer = self._adam(self.model.parameters(), lr=args.learning_rate, weight_decay=args.weight_decay)
        self._graphed_steps = {}

    def set_ft_optimizer(self, args):
        if args.model == 'pinnsf_res':
            corr = list(self.model.corrector.parameters())
            ids = {id(p) for p in corr}
            rest = [p for p in self.model.parameters() if id(p) not in ids]
            self.optimizer = self._adam(
                [{'params': corr, 'lr': args.learning_rate * args.ft_lr_decay2},
                 {'params': rest, 'lr': args.learning_rate * args.finetune_lr_decay}],
                lr=args.learning_rate, weight_decay=args.weight_decay)
        else:
            self.optimizer = self._adam(self.model.parameters(), lr=args.learning_rate * args.finetune_lr_decay,
                                        weight_decay=args.weight_decay * args.finetune_wd_aug)
        self._graphed_steps = {}

    def set_scheduler(self, args):
        self.scheduler = None

    set_ft_scheduler = set_scheduler

    # ---- losses (simulators.py:141-249) ----
    @staticmethod
    def get_dist(pred, label):
        return torch.norm(pred - label, p=2, dim=-1)

    @staticmethod
    def reduction(values, mode):
        if mode == 'sum':
            return torch.sum(values)
        if mode == 'mean':
            return torch.mean(values)
        if mode == 'none':
            return values
        raise NotImplementedError

    def loss_func(self, pred, labels, reduction='none'):
        return F.mse_loss(pred, labels, reduction=reduction)

    def l1_reg_loss(self, embeddings, weight=1e-3, reduction='none'):
        return self.reduction(weight * torch.abs(embeddings), reduction)

    def multiple_rollout_mse_loss(self, pred, labels, time_decay, reduction='none', reverse=False):
        """(c,t,n,2) squared error weighted by time_decay^(T-1-t) (or ^t when `reverse`)."""
        T = pred.shape[1]
        steps = torch.arange(T, device=pred.device, dtype=pred.dtype)
        expo = steps if reverse else (T - 1 - steps)
        decay = torch.pow(float(time_decay), expo).reshape(1, T, 1, 1)        # scalar base: no host-to-device copy
        return self.reduction((pred - labels) ** 2 * decay, reduction)

    def multiple_rollout_collision_avoidance_loss(self, pred, labels, time_decay, reduction='none'):
        """MSE of the components perpendicular to each agent's label displacement."""
        ni = labels[:, -1:, :, :] - labels[:, 0:1, :, :]
        ni = ni / (torch.norm(ni, p=2, dim=-1, keepdim=True) + 1e-6)
        pred_ = pred - torch.sum(pred * ni, dim=-1, keepdim=True) * ni
        labels_ = labels - torch.sum(labels * ni, dim=-1, keepdim=True) * ni
        return self.reduction(self.multiple_rollout_mse_loss(pred_, labels_, time_decay, reduction='none'), reduction)

    def multiple_rollout_collision_loss(self, pred, labels, time_decay, coll_focus_weight, collisions,
                                        reduction='none', abnormal_mask=None):
        """Collision-focus loss: the avoidance loss of agents that collided anywhere in the window."""
        w = (torch.sum(collisions, dim=1) > 0).to(pred.dtype)                 # c, n
        w = w.unsqueeze(1).unsqueeze(-1)                                      # c, 1, n, 1
        loss = w * self.multiple_rollout_collision_avoidance_loss(pred, labels, time_decay, reduction='none')
        if abnormal_mask is not None:
            loss = loss * abnormal_mask.reshape(1, 1, -1, 1)
        return self.reduction(loss, reduction)

    # ---- checkpoints (simulators.py:254-289) ----
    # The reference always writes ../saved_model/{exp_name}_{suffix}[_finetuned] and re-reads it (fine-tuning starts
    # from the best pre-trained weights, every evaluation loads the best ones).  Here the best weights of each stage
    # are always snapshotted in memory (`self._checkpoints`), and additionally written to `--save_dir` when given.
    def _ckpt_path(self, args, finetune_flag):
        path = os.path.join(getattr(args, 'save_dir', '') or '../saved_model', f'{args.exp_name}_{args.model_name_suffix}')
        return path + ('_finetuned' if finetune_flag else '')

    def load_model(self, args, set_model=True, finetune_flag=True, load_path=''):
        if set_model:
            (self.set_ft_model if finetune_flag else self.set_model)(args)
        snap = self._checkpoints.get(bool(finetune_flag))
        if load_path or (snap is None and getattr(args, 'save_dir', None)):
            sd = torch.load(load_path or self._ckpt_path(args, finetune_flag), map_location=args.device)
        elif snap is not None:
            sd = snap
        else:
            raise FileNotFoundError('load_model: no checkpoint of the {} stage (nothing saved yet and no --save_dir)'
                                    .format('fine-tune' if finetune_flag else 'pre-train'))
        if next(iter(sd)).startswith('module.'):           # saved from nn.DataParallel
            sd = {k[7:]: v for k, v in sd.items()}
        self.model.load_state_dict(sd)                     # in place: captured graphs keep reading the same buffers

    def save_model(self, args, finetune_flag=True, cpu_version=False):
        sd = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        self._checkpoints[bool(finetune_flag)] = sd
        if getattr(args, 'save_dir', None):
            path = self._ckpt_path(args, finetune_flag) + ('_cpu' if cpu_version else '')
            os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
            torch.save({k: v.cpu() for k, v in sd.items()} if cpu_version else sd, path)

    # ---- shared per-step pieces ----
    def _features(self, p_cur, v_cur, a_cur, dest_cur, obstacles):
        """get_relative_features on one frame of state (simulators.py:642-649 / 772-776)."""
        a = self.args
        pf, of, df = self.get_relative_features(
            p_cur.unsqueeze(-3), v_cur.unsqueeze(-3), a_cur.unsqueeze(-3), dest_cur.unsqueeze(-3), obstacles,
            a.topk_ped, a.sight_angle_ped, a.dist_threshold_ped, a.topk_obs, a.sight_angle_obs,
            a.dist_threshold_obs)
        return pf.squeeze(-4), of.squeeze(-4), df.squeeze(-3)

    @staticmethod
    def _inject(new_mask, cur, truth):
        """cur[new] = truth[new] without a host-side nonzero()."""
        m = new_mask.unsqueeze(-1) if cur.dim() > new_mask.dim() else new_mask
        return torch.where(m, truth, cur)

    def _packed_weights(self):
        """`model.packed_weights()` (one weight pack for every forward pass inside the block) where the model has it."""
        fn = getattr(self.model, 'packed_weights', None)
        return fn() if fn is not None else contextlib.nullcontext()

    # ---- HOT LOOP B: inference rollout (simulators.py:556-657) ----
    def _rollout_state(self, data, t_start):
        """Persistent (static-address) buffers of one rollout; the frame counter lives on the
        device so that one captured step can be replayed for every frame."""
        st = types.SimpleNamespace()
        dev = data.position.device
        st.td = data.position.dim() - 3                     # time axis of (*c, t, n, 2) tensors
        st.T = data.num_frames
        st.pf = data.ped_features[..., t_start, :, :, :].clone()
        st.of = data.obs_features[..., t_start, :, :, :].clone()
        st.selff = data.self_features[..., t_start, :, :].clone()
        st.desired_speed = st.selff[..., -1:].clone()
        st.hist = st.selff[..., 2:-3].clone()               # *c, n, 2*num_history_velocity
        st.a = data.acceleration[..., t_start, :, :].clone()
        st.v = data.velocity[..., t_start, :, :].clone()
        st.p = data.position[..., t_start, :, :].clone()
        st.dest = data.destination[..., t_start, :, :].clone()
        st.dest_idx = data.dest_idx[..., t_start, :].clone()
        st.p_res = torch.zeros_like(data.position)
        st.v_res = torch.zeros_like(data.velocity)
        st.a_res = torch.zeros_like(data.acceleration)
        st.p_res[..., :t_start + 1, :, :] = data.position[..., :t_start + 1, :, :]
        st.v_res[..., :t_start + 1, :, :] = data.velocity[..., :t_start + 1, :, :]
        st.a_res[..., :t_start + 1, :, :] = data.acceleration[..., :t_start + 1, :, :]
        st.mask_new = torch.zeros_like(data.mask_p_pred, dtype=torch.float32)
        st.mask_new[..., :t_start + 1, :] = data.mask_p[..., :t_start + 1, :].long()
        new_flag = (data.mask_p - data.mask_p_pred).long() == 1                       # *c, t, n
        pad = torch.zeros_like(new_flag.narrow(st.td, 0, 1))
        st.new_flag = torch.cat((new_flag, pad), dim=st.td)                          # frame T: nobody enters
        st.t = torch.full((1,), t_start, device=dev, dtype=torch.long)
        st.nan = torch.tensor(float('nan'), device=dev)
        # contiguous views / copies the fused epilogue kernel indexes with the device-side frame counter
        st.new_flag_u8 = st.new_flag.to(torch.uint8).contiguous()
        st.series = {k: getattr(data, k).contiguous() for k in
                     ('position', 'velocity', 'acceleration', 'destination', 'self_features')}
        st.series['dest_idx'] = data.dest_idx.long().contiguous()
        st.waypoints = data.waypoints.contiguous()
        st.dest_num = data.dest_num.long().contiguous().to(dev)
        st.dest_idx = st.dest_idx.long().contiguous()
        st.ped_idx = torch.empty(st.pf.shape[:-1], device=dev, dtype=torch.int32)
        st.obs_idx = torch.empty(st.of.shape[:-1], device=dev, dtype=torch.int32)
        for k in ('pf', 'of', 'selff', 'hist', 'p', 'v', 'a', 'dest', 'desired_speed'):
            setattr(st, k, getattr(st, k).contiguous())
        return st

    def _rollout_step_fused(self, data, st):
        """The same frame as `_rollout_step` in three launches besides the MLP: the fused integrator
        epilogue (piml_rollout_step), the relative-feature kernel writing straight into the state
        buffers (which also advances the frame counter)."""
        a = self.args
        # the bottleneck variants leave their epilogue (neighbour-axis sums + desired force) to the integrator's launch
        defer = STEP_WITH_KSUM and st.selff.dim() == 2 and hasattr(self.model, 'defer_ksum_epilogue')
        if defer:
            self.model.defer_ksum_epilogue, self.model.pending_ksum = True, None
        try:
            a_next = self.model(st.pf, st.of, st.selff)[0]
            ksum = getattr(self.model, 'pending_ksum', None) if defer else None
        finally:
            if defer:
                self.model.defer_ksum_epilogue, self.model.pending_ksum = False, None
        if a_next is None:
            ops.rollout_step(st, data, None, remove_arrived=True, ksum=ksum)
        else:
            ops.rollout_step(st, data, a_next.contiguous(), remove_arrived=True)
        ops.relative_features_into((st.pf, st.of, st.selff, st.ped_idx, st.obs_idx), st.p, st.v, st.a, st.dest,
                                   data.obstacles, a.topk_ped, a.sight_angle_ped, a.dist_threshold_ped,
                                   a.topk_obs, a.sight_angle_obs, a.dist_threshold_obs, tick=st.t)   # + st.t += 1

    def _rollout_step(self, data, st):
        """One simulated frame (the body of simulators.py:595-652) on the persistent buffers.
        Every index into the time axis comes from the device-side counter `st.t`."""
        dt = data.time_unit
        td, t = st.td, st.t
        st.p_res.index_copy_(td, t, st.p.unsqueeze(td))
        st.v_res.index_copy_(td, t, st.v.unsqueeze(td))
        st.a_res.index_copy_(td, t, st.a.unsqueeze(td))
        row = st.mask_new.index_select(td, t).squeeze(td)
        st.mask_new.index_copy_(td, t, torch.where(st.p[..., 0].isnan(), row, torch.ones_like(row)).unsqueeze(td))

        a_next = self.model(st.pf, st.of, st.selff)[0]                        # :602
        v_next = st.v + st.a * dt                                             # lagged Euler, quirk Q6
        p_next = st.p + st.v * dt

        near = torch.norm(st.p - st.dest, p=2, dim=-1) < 0.5                  # :608-609
        dest_idx = st.dest_idx + near.long()
        gone = dest_idx > data.dest_num - 1                                   # arrived at the last waypoint
        p_next = torch.where(gone.unsqueeze(-1), st.nan, p_next)              # leaves the scene (:611)
        dest_idx = dest_idx - gone.long()
        dest = _gather_waypoints(data.waypoints, dest_idx)
        hist = torch.cat((st.hist[..., 2:], v_next), dim=-1)                  # :624-626

        # newly entering agents are injected from the ground truth of frame t+1 (:629-639)
        tn = t + 1
        tc = torch.clamp(tn, max=st.T - 1)
        new = st.new_flag.index_select(td, tn).squeeze(td)

        def frame(x):
            return x.index_select(td, tc).squeeze(td)
        p_next = self._inject(new, p_next, frame(data.position))
        v_next = self._inject(new, v_next, frame(data.velocity))
        a_next = self._inject(new, a_next, frame(data.acceleration))
        dest = self._inject(new, dest, frame(data.destination))
        dest_idx = self._inject(new, dest_idx, frame(data.dest_idx))
        hist = self._inject(new, hist, frame(data.self_features)[..., 2:-3])

        pf, of, df = self._features(p_next, v_next, a_next, dest, data.obstacles)
        st.p.copy_(p_next); st.v.copy_(v_next); st.a.copy_(a_next)
        st.dest.copy_(dest); st.dest_idx.copy_(dest_idx); st.hist.copy_(hist)
        st.pf.copy_(pf); st.of.copy_(of)
        st.selff.copy_(torch.cat((df, hist, a_next, st.desired_speed), dim=-1))     # :651
        st.t.add_(1)

    def get_multiple_rollouts(self, data, t_start=0, load_model=True, use_graph=None, fused=None):
        """Roll the scene forward from frame `t_start` (simulators.py:556-657).  On the GPU, under
        no_grad, the per-frame body is captured once into a HIP graph and replayed for the
        remaining frames (`use_graph=None` = automatic): the ~60 small launches of a frame become
        one graph launch."""
        args = self.args
        if load_model:
            self.load_model(args, set_model=False, finetune_flag=self.finetune_flag)
        # inference frames read the predicted accelerations only: PINNSF_multitask's auxiliary collision head is not launched
        # (set before the weights are packed: the pack then holds no head either)
        only = not torch.is_grad_enabled() and hasattr(self.model, 'predictions_only')
        before = getattr(self.model, 'predictions_only', False)
        if only:
            self.model.predictions_only = True
        try:
            with self._packed_weights():      # the weights do not change during a rollout: packed once, not per frame
                return self._multiple_rollouts(data, t_start, use_graph, fused)
        finally:
            if only:
                self.model.predictions_only = before

    def _multiple_rollouts(self, data, t_start, use_graph, fused):
        args = self.args
        st = self._rollout_state(data, t_start)
        steps = st.T - t_start
        if fused is None:
            fused = data.position.is_cuda and not torch.is_grad_enabled()
        step_fn = self._rollout_step_fused if fused else self._rollout_step
        if use_graph is None:
            use_graph = data.position.is_cuda and not torch.is_grad_enabled() and steps > 8
        from .. import hip_graphs_safe
        use_graph = bool(use_graph) and hip_graphs_safe()       # a process that cannot trust replays runs eagerly
        done = 0
        quirk = bool(getattr(args, 'inplace_quirk', False)) and steps > 0
        hist_first = None
        if quirk:                                         # the first frame on its own: its history is carried over
            step_fn(data, st)
            hist_first = st.hist.clone()
            done = 1
        if use_graph and steps - done > 3:
            try:
                if self._side_stream_ok(st.p.numel() // 2):   # obstacle branch in parallel inside the graph
                    self.model.obs_stream = torch.cuda.Stream()
                for _ in range(2):                        # real frames, also warm every lazy init up
                    step_fn(data, st)
                done += 2
                torch.cuda.synchronize()
                # frames per graph (the frames are identical: device-side counter).  One is the default: eight frames per
                # launch were measured SLOWER (52 vs 46 us/frame at N = 122, 108 vs 99 at N = 4096) -- the frame is GPU-bound
                # (40 us of kernels + ~1 us between dependent launches), not bound by the host's graph launches
                per = max(1, min(int(getattr(args, 'frames_per_graph', 1)), steps - done))
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for _ in range(per):
                        step_fn(data, st)
                self.model.obs_stream = None
                done += per                                # the capture does not execute: replay it once now
                graph.replay()
                for _ in range((steps - done) // per):
                    graph.replay()
                done += (steps - done) // per * per
            except RuntimeError as ex:                    # capture unsupported: finish eagerly
                self.model.obs_stream = None
                print(f'[piml_amd] rollout graph capture failed ({ex}); continuing eagerly')
                torch.cuda.synchronize()
                done = int(st.t.item()) - t_start
        for _ in range(steps - done):
            step_fn(data, st)
        if quirk:                                         # quirk Q12: what the reference's views leave behind
            with torch.no_grad():
                data.dest_idx[..., t_start, :] = st.dest_idx.to(data.dest_idx.dtype)
                data.self_features[..., t_start, :, 2:-3] = hist_first
        return RolloutResult(position=st.p_res, velocity=st.v_res, acceleration=st.a_res,
                             destination=data.destination, waypoints=data.destination, obstacles=data.obstacles,
                             mask_p=st.mask_new, meta_data=getattr(data, 'meta_data', None),
                             time_unit=data.time_unit)

    # ---- HOT LOOP C: differentiable rollout for fine-tuning (simulators.py:659-832) ----
    def test_multiple_rollouts_for_training(self, data, t_start=0):
        """(loss, mse_loss, collision_loss, hard_collision_loss, collision_pred_loss, collision_pred_acc,
        reg_loss) of one batch of rollout windows, differentiable (simulators.py:659-832)."""
        out, aux = self._training_rollout(data, t_start)
        assert not bool(aux['nan_seen']), f'find nan in epoch : {self.epoch} {self.batch_idx}'  # :745
        self._carry_dest_idx(data, aux)
        self.collision_count += aux['collisions'].item()
        self.hard_collision_count += aux['hard_collisions'].item()
        return out

    def _carry_dest_idx(self, data, aux):
        """quirk Q12 for the training rollout: `dest_idx_cur` is a view of data.dest_idx[..., t_start, :] in the
        reference (simulators.py:684, 749-751, 768), so a batch object re-used in the next epoch starts from the
        waypoint indices its previous rollout ended with."""
        if getattr(self.args, 'inplace_quirk', False):
            with torch.no_grad():
                data.dest_idx[..., aux['t_start'], :] = aux['dest_idx_final'].to(data.dest_idx.dtype)

    def _training_rollout(self, data, t_start=0):
        """The rollout + losses with no host synchronisation at all (capturable into a HIP graph);
        returns the 7 loss tensors and the device-side bookkeeping (NaN flag, collision totals)."""
        with self._packed_weights():          # one weight pack for all frames of the window (the optimizer steps later)
            return self._training_rollout_frames(data, t_start)

    def _training_rollout_frames(self, data, t_start):
        args = self.args
        dt = data.time_unit
        waypoints, obstacles, dest_num = data.waypoints, data.obstacles, data.dest_num
        T = data.num_frames
        labels = data.labels
        thr = args.collision_threshold

        state = [data.ped_features[..., t_start, :, :, :], data.obs_features[..., t_start, :, :, :],
                 data.self_features[..., t_start, :, :]]
        # channelled (C, T, N, .) batches on the fused frame step: the clones of frame t_start, the new-pedestrian flags, the
        # integer mask, the per-frame gates and the desired speeds in ONE launch (ops.rollout_prologue; fourteen otherwise)
        pro = None
        if self.fused_train_step and data.position.dim() == 4 and data.position.is_cuda and T == data.position.shape[1]:
            pro = ops.rollout_prologue(data, t_start)
        if pro is not None and state[0].dim() == 4 and not state[0].is_contiguous():
            # frame t_start of the three feature arrays, contiguous, in ONE launch (C x 3 contiguous chunks: ops.multi_copy) -- the
            # fused network made each of them contiguous with a strided copy of its own
            C0 = state[0].shape[0]
            bufs = [torch.empty(tuple(x.shape), device=x.device, dtype=x.dtype) for x in state]
            ops.multi_copy([b[c] for b in bufs for c in range(C0)], [x[c] for x in state for c in range(C0)])
            state = bufs
        if pro is not None:
            mask_pred = pro['mask_pred']
            desired_speed = pro['speed']
            a_cur, v_cur, p_cur, dest_cur, dest_idx = pro['a'], pro['v'], pro['p'], pro['dest'], pro['dest_idx']
            new_flag = None
        else:
            mask_pred = data.mask_p_pred.long()                               # c, t, n   (read only: no copies of it or of the labels)
            desired_speed = state[2][..., -1:]
            a_cur = data.acceleration[..., t_start, :, :].clone()
            v_cur = data.velocity[..., t_start, :, :].clone()
            p_cur = data.position[..., t_start, :, :].clone()
            dest_cur = data.destination[..., t_start, :, :].clone()
            dest_idx = data.dest_idx[..., t_start, :].clone()
            new_flag = (data.mask_p - data.mask_p_pred).long() == 1

        dev = p_cur.device
        p_steps, a_steps, cnt_steps, lab_steps = [], [], [], []
        bm_head = args.collision_pred_weight > 0 and args.model == 'pinnsf_bm'
        zero = self._const(dev, 0.0)                # (one persistent zero for every sum that starts at 0: each torch.zeros is a launch)
        pred_collisions = true_collision = zero
        pred_steps, true_steps = [], []             # bm head: per-frame records, stacked and gated once after the loop
        fused_bce = bm_head and self.fused_rollout_losses and p_cur.is_cuda and p_cur.dtype == torch.float32 and T - t_start <= 32
        loss = zero
        reg_loss = zero
        nan_seen = None                             # (the fused frame step keeps its own flag on the device)
        # `if torch.sum(mask) > 0` of every frame (:707), evaluated once for all frames: the per-frame records
        # below are gated after the loop in one pass instead of frame by frame
        if pro is not None:
            gates, gates_f = pro['gates'], pro['gates_f']
        else:
            gates = mask_pred.sum(dim=(0, 2)) > 0                             # (T,)
            gates_f = gates.to(p_cur.dtype)
        need_label_counts = bool(args.new_collision_loss_flag)                # label collisions are only read there
        if need_label_counts:
            lab_frames = labels[..., :2].transpose(0, 1).contiguous()         # (T, C, N, 2): frame t is contiguous
        # channelled (C, T, N, .) batches take the fused frame step (piml_train_step_fwd/bwd)
        fused_step = self.fused_train_step and p_cur.dim() == 3 and p_cur.is_cuda
        if fused_step:
            series = tuple(x.contiguous() for x in (data.position, data.velocity, data.acceleration,
                                                    data.destination, data.dest_idx.long()))
            dest_num_i64 = dest_num.long().to(dev).contiguous()
            if pro is not None:
                new_flag_u8, nan_flag, speed_rows = pro['new_flag_u8'], pro['nan_flag'], pro['speed']
            else:
                new_flag_u8 = new_flag.contiguous().view(torch.uint8)
                dest_idx = dest_idx.long()
                nan_flag = torch.zeros((), device=dev, dtype=torch.int32)
                speed_rows = desired_speed.contiguous()

        batch_counts = BATCH_COUNTS and p_cur.is_cuda and p_cur.dim() == 3 and p_cur.shape[0] <= 25 and T - t_start <= 32
        # the (C, T, N, 2) array of predicted positions the loss reads, filled by the frame steps themselves (each writes its INPUT
        # position into its frame: ops.rollout_frame stack=) instead of by a concatenation behind the loop
        p_buf = None
        if fused_step and pro is not None and t_start == 0 and not ops.DETERMINISTIC_BWD:
            p_buf = torch.empty(p_cur.shape[0], T, p_cur.shape[1], 2, device=dev, dtype=torch.float32)
        # the model's tail under the agent-axis norm rides in the frame step's launch (ops.rollout_frame tail=; forward and backward)
        park_tail = fused_step and self.tail_in_step and hasattr(self.model, 'defer_train_tail') and not ops.DETERMINISTIC_BWD
        for t in range(t_start, T):
            tail = None
            if park_tail:
                self.model.defer_train_tail, self.model.pending_tail = True, None
            try:
                predictions = self.model(*state)                              # :701
            finally:
                if park_tail:
                    tail = self.model.pending_tail
                    self.model.defer_train_tail, self.model.pending_tail = False, None
            p_msg = predictions[1]
            gf = gates_f[t]

            if not batch_counts:
                cnt_steps.append(ops.collision_counts(p_cur, (thr, thr / 2)))     # :708-715, fused, quirk Q7
            if need_label_counts:
                lab_steps.append(ops.collision_counts(lab_frames[t], (thr, thr / 2)))   # :717-724
            p_steps.append(p_cur)                                             # :728-729
            a_steps.append(a_cur)
            if bm_head:                                                       # :731-733
                pred_steps.append(predictions[-1])
                # (fused: the label is evaluated inside the loss launch, from the frame's pedestrian features where they lie)
                true_steps.append(state[0] if fused_bce else self.calculate_collision_label(state[0]))
            if args.reg_weight > 0:                                           # :735-737 (cumulative, as shipped)
                reg_loss = reg_loss + self.l1_reg_loss(p_msg, args.reg_weight, 'sum') * gf
                loss = loss + reg_loss * gf

            a_next = predictions[0]
            if fused_step:
                # :741-769 as one differentiable launch (integrator, waypoint switch, injection, NaN flag) and :772-779 (the
                # features of the new state incl. the self_features rows) as a second, both inside ONE autograd node
                # (the frame's input position comes back as an ALIAS output: the loss reads that, so the tensor has one consumer and
                # the loss's gradient is added inside the frame's backward launch -- no strided addition of the engine per frame)
                p_cur, v_cur, a_cur, dest_cur, dest_idx, *state, p_alias = ops.rollout_frame(
                    p_cur, v_cur, a_cur, a_next, dest_cur, dest_idx, waypoints, dest_num_i64, dt, new_flag_u8, series, t + 1,
                    nan_flag, obstacles, speed_rows, args.topk_ped, args.sight_angle_ped, args.dist_threshold_ped,
                    args.topk_obs, args.sight_angle_obs, args.dist_threshold_obs, alias_position=True,
                    stack=None if p_buf is None else (p_buf, t), tail=tail)
                p_steps[-1] = p_alias
            else:
                nan_seen = a_next.isnan().any() if nan_seen is None else (nan_seen | a_next.isnan().any())
                v_next = v_cur + a_cur * dt                                   # :741-743
                p_next = p_cur + v_cur * dt

                near = torch.norm(p_cur - dest_cur, p=2, dim=-1) < 0.5       # :748-754, nobody is removed
                dest_idx = dest_idx + near.long()
                dest_idx = dest_idx - (dest_idx > dest_num - 1).long()
                dest_cur = _gather_waypoints(waypoints, dest_idx)
                p_cur, v_cur, a_cur = p_next, v_next, a_next

                if t < T - 1:                                                 # :762-769
                    new = new_flag[..., t + 1, :]
                    p_cur = self._inject(new, p_cur, data.position[..., t + 1, :, :])
                    v_cur = self._inject(new, v_cur, data.velocity[..., t + 1, :, :])
                    a_cur = self._inject(new, a_cur, data.acceleration[..., t + 1, :, :])
                    dest_cur = self._inject(new, dest_cur, data.destination[..., t + 1, :, :])
                    dest_idx = self._inject(new, dest_idx, data.dest_idx[..., t + 1, :])

            if not fused_step:
                pf, of, df = self._features(p_cur, v_cur, a_cur, dest_cur, obstacles)   # :772-776, differentiable
                state = [pf, of, torch.cat((df, v_cur, a_cur, desired_speed), dim=-1)]  # :778-779

        if batch_counts:        # :708-715 for every frame of the window in one launch (the positions are not differentiated here)
            cnt_steps = ops.collision_counts_frames(p_steps, (thr, thr / 2))

        def frames(steps):
            """per-frame (2, C, N) count records -> two gated (C, T, N) tensors"""
            pad = [torch.zeros_like(steps[0])] * t_start if t_start else []
            allf = torch.stack(pad + steps, dim=2) * gates_f.view(1, 1, -1, 1)    # (2, C, T, N)
            return allf[0], allf[1]
        # the frames' count records straight into the loss launch (ops.rollout_losses_frames: gated there, the collision totals
        # and the predicted-entry count the step logs come back with the sums, the loss weights are applied inside): no stack,
        # no products, no separate reductions
        use_frames = (self.fused_rollout_losses and not need_label_counts and T <= 32 and p_steps[0].is_cuda
                      and p_steps[0].dim() == 3 and p_steps[0].dtype == torch.float32)
        collisions = hard_collisions = None
        if not use_frames:
            collisions, hard_collisions = frames(cnt_steps)
        if need_label_counts:                                                 # :782-788
            label_collisions, label_hard = frames(lab_steps)
            collisions = collisions * (label_collisions.sum(dim=-2, keepdim=True) <= 0)
            hard_collisions = hard_collisions * (label_hard.sum(dim=-2, keepdim=True) <= 0)
        aux = {'t_start': t_start}
        if fused_step and pro is not None:
            # the captured step reads the raw flag (BaseSimulator._pack_log_vec); `!= 0` is a launch, and so is a clone of the last
            # frame's waypoint indices -- a fresh tensor of the frame step that nothing writes again
            aux['nan_flag_i32'] = nan_flag
            nan_seen = nan_flag
            aux['dest_idx_final'] = dest_idx.detach()
        else:
            if fused_step:
                nan_seen = nan_flag != 0
            aux['dest_idx_final'] = dest_idx.detach().clone()
        aux['nan_seen'] = nan_seen
        if not use_frames:
            aux['collisions'], aux['hard_collisions'] = torch.sum(collisions), torch.sum(hard_collisions)

        pad = [torch.zeros_like(p_steps[0])] * t_start if t_start else []
        gate4 = gates.view(1, -1, 1, 1)
        p_res = ops.stack_of(p_buf, 0, p_steps) if p_buf is not None else torch.stack(pad + p_steps, dim=1)     # c, t, n, 2
        collision_loss, hard_collision_loss, collision_pred_loss, collision_pred_acc = zero, zero, zero, zero
        want_coll = args.collision_loss_weight > 0 and args.collision_loss_version in ('v0', 'v2')
        fused_losses = self.fused_rollout_losses and p_res.is_cuda and p_res.dim() == 4 and p_res.dtype == torch.float32
        if fused_losses and use_frames:
            am = data.abnormal_mask if args.collision_loss_version == 'v2' else None
            w_c = args.collision_loss_weight if want_coll else 0.0
            total, mse_loss, coll_w, hard_w, stats = ops.rollout_losses_frames(
                p_res, labels, mask_pred, gates, [None] * t_start + cnt_steps, want_coll, am, args.time_decay, w_c,
                w_c * args.hard_collision_penalty)
            loss = total if loss is zero else loss + total
            if want_coll:
                collision_loss, hard_collision_loss = coll_w, hard_w
            aux['collisions'], aux['hard_collisions'], aux['n_pred'] = stats[0], stats[1], stats[2]
            if args.teacher_weight > 0:
                labels = torch.where((mask_pred != 0).unsqueeze(-1), labels, torch.zeros_like(labels))   # :794
        elif fused_losses:
            # :790-819 as ONE launch forward and one backward (ops.rollout_losses: the masks, the time-decayed squared error
            # and the two collision-focus sums; on torch operators ~100 launches of a few microseconds each)
            am = data.abnormal_mask if args.collision_loss_version == 'v2' else None
            sums = ops.rollout_losses(p_res, labels, mask_pred, gates, collisions if want_coll else None,
                                      hard_collisions if want_coll else None, am, args.time_decay)
            mse_loss = sums[0]
            loss = loss + mse_loss
            if want_coll:
                collision_loss = sums[1] * args.collision_loss_weight
                hard_collision_loss = sums[2] * (args.collision_loss_weight * args.hard_collision_penalty)
                loss = loss + collision_loss + hard_collision_loss
            if args.teacher_weight > 0:
                labels = torch.where((mask_pred != 0).unsqueeze(-1), labels, torch.zeros_like(labels))   # :794
        else:
            keep = (mask_pred != 0).unsqueeze(-1)
            p_res = torch.where(keep & gate4, p_res, torch.zeros_like(p_res))     # :728 gate and :793 delete 'nan'
            labels = torch.where(keep, labels, torch.zeros_like(labels))          # :794
            lab_p = labels[:, :, :, :2]
            mse_loss = self.multiple_rollout_mse_loss(p_res, lab_p, args.time_decay, reduction='sum')
            loss = loss + mse_loss
        if not fused_losses and want_coll:                                    # :800-819
            am = data.abnormal_mask if args.collision_loss_version == 'v2' else None
            collision_loss = self.multiple_rollout_collision_loss(
                p_res, lab_p, args.time_decay, args.collision_focus_weight, collisions, reduction='sum',
                abnormal_mask=am) * args.collision_loss_weight
            hard_collision_loss = self.multiple_rollout_collision_loss(
                p_res, lab_p, args.time_decay, args.collision_focus_weight, hard_collisions, reduction='sum',
                abnormal_mask=am) * args.collision_loss_weight * args.hard_collision_penalty
            loss = loss + collision_loss + hard_collision_loss
        if args.teacher_weight > 0:                                           # :821-824
            a_res = torch.stack(pad + a_steps, dim=1)
            a_res = torch.where(gate4, a_res, torch.zeros_like(a_res))        # :729 (NaN-safe gate)
            a_mse = self.multiple_rollout_mse_loss(a_res, labels[..., 4:6], args.time_decay, reduction='sum',
                                                   reverse=True)
            loss = loss + a_mse * args.teacher_weight
        if args.collision_pred_weight > 0 and not bm_head:
            # :826-830 with both tensors still all zeros (only `pinnsf_bm` fills them, :731-733): BCE(0, 0) = 0 and every
            # rounded prediction equals its label -- the values the six launches below would compute
            collision_pred_acc = self._const(dev, 1.0)
        elif args.collision_pred_weight > 0 and fused_bce:                    # :826-830 as one launch each way
            collision_pred_loss, collision_pred_acc = ops.collision_pred_loss(pred_steps, true_steps, gates_f, t_start, T,
                                                                              args.collision_pred_weight)
            loss = loss + collision_pred_loss
        elif args.collision_pred_weight > 0:                                  # :826-830
            # (c, t, n, k) records of :731-733: zeros in the frames before t_start, every frame times its gate -- one stack and
            # one product instead of two slice assignments per frame
            gk = gates_f.view(1, -1, 1, 1)
            padk = [torch.zeros_like(true_steps[0])] * t_start if t_start else []
            pred_collisions = torch.stack(padk + pred_steps, dim=1) * gk
            true_collision = torch.stack(padk + true_steps, dim=1) * gk
            collision_pred_loss = F.binary_cross_entropy(pred_collisions, true_collision,
                                                         reduction='sum') * args.collision_pred_weight
            collision_pred_acc = torch.sum(torch.round(pred_collisions) == true_collision) / true_collision.numel()
            loss = loss + collision_pred_loss
        return (loss, mse_loss, collision_loss, hard_collision_loss, collision_pred_loss, collision_pred_acc,
                reg_loss), aux

    # ---- pointwise evaluation and the Adam loop (simulators.py:291-440) ----
    def test_pointwise(self, data):
        self.model.eval()
        with torch.no_grad():
            ped_features, obs_features, self_features, labels = data[:]
            pred = self.model(ped_features, obs_features, self_features)[0]
            loss = torch.mean(self.loss_func(pred, labels[:, 4:6])).item()
        return loss, loss

    _BATCH_TENSORS = ('ped_features', 'obs_features', 'self_features', 'labels', 'mask_p', 'mask_p_pred', 'position',
                      'velocity', 'acceleration', 'destination', 'dest_idx', 'waypoints', 'dest_num', 'obstacles',
                      'abnormal_mask')

    def _graphed_rollout_step(self, batch):
        """Whole fine-tuning step (zero_grad, T-frame differentiable rollout, losses, backward, Adam)
        replayed from ONE captured HIP graph.  One graph per batch geometry; the batch is copied into
        the graph's static input buffers.  Weights / Adam state touched by the warm-up iterations are
        restored before capture, so training is step-for-step the eager sequence."""
        key = tuple((k, tuple(getattr(batch, k).shape)) for k in self._BATCH_TENSORS) + (id(self.optimizer), self.model.training) + self._path_flags()
        entry = self._graphed_steps.get(key)
        if entry is None:
            # train mode: the processors' dropout masks are drawn on the device from a (seed, call counter) state that
            # the draws advance themselves (ops.dropout_keep_bits), so every replay sees fresh masks; the counter the
            # warm-up iterations advanced is put back, like the weights
            drop_state = ops.dropout_state(batch.position.device)
            drop_calls = drop_state[1].clone()
            static = types.SimpleNamespace(**{k: v for k, v in batch.__dict__.items() if not isinstance(v, torch.Tensor)})
            for k in self._BATCH_TENSORS:
                setattr(static, k, getattr(batch, k).clone())
            params = [p for g in self.optimizer.param_groups for p in g['params']]
            saved_p = [p.detach().clone() for p in params]
            saved_s = {id(p): {k: v.clone() for k, v in self.optimizer.state.get(p, {}).items() if torch.is_tensor(v)}
                       for p in params}

            one = self._const(static.position.device, 1.0)
            self._const(static.position.device, 0.0)

            def one_step():
                self.optimizer.zero_grad(set_to_none=True)
                # the frames' weight gradients summed inside the slot-sum launches (the sink), and those launches riding as the
                # leading workgroups of each frame's relfeat backward (ops.deferred_slot_sums: one launch less per frame)
                with self._grad_sink.step(), ops.deferred_slot_sums():
                    out, aux = self._training_rollout(static)
                    out[0].backward(gradient=one)                 # (the persistent one: no ones_like fill in the graph)
                self.optimizer.step()
                return out, aux
            # inside the graph the obstacle branch of the MLP runs on a side stream (parallel chains of
            # small kernels); eager execution keeps a single stream
            if self._side_stream_ok(static.position.numel() // 2 // static.position.shape[1]):
                self.model.obs_stream = torch.cuda.Stream()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    one_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            with torch.no_grad():                         # undo the warm-up updates, in place
                for p, sp in zip(params, saved_p):
                    p.copy_(sp)
                    for k, v in self.optimizer.state.get(p, {}).items():
                        if torch.is_tensor(v):
                            v.copy_(saved_s[id(p)][k]) if k in saved_s[id(p)] else v.zero_()
                drop_state[1].copy_(drop_calls)
            self.optimizer.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out, aux = one_step()
                # every scalar the host reads after the step in ONE vector (one device-to-host read instead of eleven
                # synchronising ones): the 7 loss terms, the two collision totals, the NaN flag, the predicted-agent count
                aux['log_vec'] = self._pack_log_vec(out, aux, static)
            self.model.obs_stream = None
            entry = (graph, static, out, aux)
            self._graphed_steps[key] = entry
        graph, static, out, aux = entry
        # the batch into the graph's static inputs: ONE launch (ops.multi_copy; torch._foreach_copy_ issued a copy per tensor)
        ops.multi_copy([getattr(static, k) for k in self._BATCH_TENSORS], [getattr(batch, k) for k in self._BATCH_TENSORS])
        graph.replay()
        return out, aux

    def _pack_log_vec(self, out, aux, static):
        """The eleven scalars the host reads after a step -- 7 loss terms, the two collision totals, the NaN flag, the predicted-agent
        count -- in ONE float32 vector.  On the fused path ONE launch (ops.multi_copy of eleven 4-byte scalars; the NaN flag travels
        as the raw int32 the frame step keeps on the device -- any set bit reads back as a non-zero float); otherwise torch.stack
        (a conversion per non-float entry, a concatenation, a copy)."""
        items = [o.detach() for o in out] + [aux['collisions'], aux['hard_collisions']]
        nan_raw = aux.get('nan_flag_i32')
        n_pred = aux.get('n_pred')
        ok = (nan_raw is not None and n_pred is not None and nan_raw.dtype == torch.int32 and
              all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.numel() == 1 for t in items + [n_pred]))
        if ok:
            vec = torch.empty(11, device=items[0].device, dtype=torch.float32)
            dsts = [vec[i:i + 1] for i in range(9)] + [vec.view(torch.int32)[9:10], vec[10:11]]
            srcs = [t.reshape(1) for t in items] + [nan_raw.reshape(1), n_pred.detach().reshape(1)]
            ops.multi_copy(dsts, srcs)
            return vec
        return torch.stack([*[o.detach().float().reshape(()) for o in out], aux['collisions'].float(), aux['hard_collisions'].float(),
                            aux['nan_seen'].float() if torch.is_tensor(aux['nan_seen']) else out[0].detach().float() * 0,
                            aux['n_pred'] if 'n_pred' in aux else (static.mask_p_pred == 1).sum().float()])

    @staticmethod
    def _path_flags():
        """The kernel-path switches a captured step depends on (a graph captured under one setting is not replayed under another)."""
        return (MODEL.FUSED_GLUE, MODEL.FUSED_ENCODER, MODEL.FUSED_NETWORK, MODEL.FUSED_ROW_DECODER, MODEL.FUSED_KSUM_TAIL)

    def _pointwise_terms(self, batch):
        """Loss terms of a pointwise batch (simulators.py:327-356): (loss, mse, reg | None, collision_pred | None)."""
        args = self.args
        ped_features, obs_features, self_features, labels = batch
        predictions = self.model(ped_features, obs_features, self_features)
        pred, p_msg = predictions[0], predictions[1]
        bm_cp = args.collision_pred_weight > 0 and args.model == 'pinnsf_bm'
        if (args.pinnsf_interaction == 'sim' and self.fused_rollout_losses and pred.is_cuda and pred.dtype == torch.float32
                and pred.dim() == 2 and labels.dim() == 2 and labels.shape[1] >= 6
                and (not bm_cp or (predictions[-1].dim() == 2 and labels.shape[1] >= 6 + predictions[-1].shape[-1]))):
            # :333-352 as ONE launch (ops.pointwise_losses: the three sums and their gradient fields)
            loss, mse_loss, reg, cp = ops.pointwise_losses(pred, labels, args.reg_weight, p_msg if args.reg_weight > 0 else None,
                                                           predictions[-1] if bm_cp else None)
            return loss, mse_loss, (reg if args.reg_weight > 0 else None), (cp if bm_cp else None)
        if args.pinnsf_interaction == 'sim':
            mse_loss = F.mse_loss(pred, labels[:, 4:6], reduction='sum')
        elif args.pinnsf_interaction == 'loss':                                        # PINN-loss pretraining
            version = 'v2' if args.iter_flag else 'v0'
            target = ops.calc_acceleration(ped_features, version, args.dataset_name)
            mse_loss = F.mse_loss(p_msg, target, reduction='sum') + \
                args.true_label_weight * F.mse_loss(pred, labels[:, 4:6], reduction='sum')
        else:
            raise NotImplementedError(args.pinnsf_interaction)
        loss, reg, cp = mse_loss, None, None
        if args.reg_weight > 0:
            reg = self.l1_reg_loss(p_msg, args.reg_weight, 'sum')
            loss = loss + reg
        if args.collision_pred_weight > 0 and args.model == 'pinnsf_bm':
            cp = F.binary_cross_entropy(predictions[-1], labels[:, 6:], reduction='sum')
            loss = loss + cp
        return loss, mse_loss, reg, cp

    def _graphed_pointwise_step(self, batch):
        """One pointwise pre-training step (HOT LOOP A: zero_grad, forward, losses, backward, Adam) replayed from ONE captured
        HIP graph per batch geometry -- eagerly the step is ~60 launches behind ~0.9 ms of Python, ctypes and autograd
        bookkeeping, whatever the batch size.  Same protocol as `_graphed_rollout_step`: the batch is copied into static
        buffers, weights / Adam state / dropout draw counter touched by the warm-up iterations are put back before the
        capture, so the sequence of updates is the eager one."""
        key = ('pointwise',) + tuple(tuple(t.shape) for t in batch) + (id(self.optimizer), self.model.training) + self._path_flags()
        entry = self._graphed_steps.get(key)
        if entry is None:
            static = [t.clone() for t in batch]
            params = [p for g in self.optimizer.param_groups for p in g['params']]
            saved_p = [p.detach().clone() for p in params]
            saved_s = {id(p): {k: v.clone() for k, v in self.optimizer.state.get(p, {}).items() if torch.is_tensor(v)}
                       for p in params}
            drop_state = ops.dropout_state(static[0].device)
            drop_calls = drop_state[1].clone()

            dev0 = static[0].device
            one, nan_c = self._const(dev0, 1.0), self._const(dev0, float('nan'))      # made outside the capture, never written

            def one_step():
                self.optimizer.zero_grad(set_to_none=True)
                # (deferred_slot_sums: the slot sums of the bottleneck variants' three operators -- head, row decoder, encoders --
                # become one launch at the block's exit; a fused network has one launch either way)
                with self._packed_weights(), ops.deferred_slot_sums():
                    terms = self._pointwise_terms(static)
                    terms[0].backward(gradient=one)               # (no ones_like fill in the graph)
                self.optimizer.step()
                return terms
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    one_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            with torch.no_grad():
                for p, sp in zip(params, saved_p):
                    p.copy_(sp)
                    for k, v in self.optimizer.state.get(p, {}).items():
                        if torch.is_tensor(v):
                            v.copy_(saved_s[id(p)][k]) if k in saved_s[id(p)] else v.zero_()
                drop_state[1].copy_(drop_calls)
            self.optimizer.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                terms = one_step()
                # the scalars the host logs, in one vector: one synchronising read per step (absent terms: NaN placeholders)
                log_vec = torch.stack([(t.detach().float().reshape(()) if t is not None else nan_c) for t in terms])
            entry = (graph, static, terms, log_vec)
            self._graphed_steps[key] = entry
        graph, static, terms, log_vec = entry
        ops.multi_copy(list(static), list(batch))
        graph.replay()
        return terms, log_vec

    def train_batch(self, batch_data):
        """One optimiser step on either batch type (the body of simulators.py:314-360).
        Returns the dict of scalar logs of this batch."""
        return self.train_batch_async(batch_data)()

    def _log_readback(self, log_vec):
        """The step's scalars on their way to the host WITHOUT waiting for them: a copy into pinned memory queued behind the step
        and an event.  Returns read() -> list of floats (waits for the event).  Two pinned buffers per vector size take turns, so
        the loop may queue the NEXT step before it reads this one's (train()) -- the captured step's own vector is overwritten
        by the next replay, the copy is ordered in front of it by the stream."""
        ring = self.__dict__.setdefault('_log_ring', {})
        key = (int(log_vec.numel()), str(log_vec.device))
        ent = ring.get(key)
        if ent is None:
            ent = ring[key] = [[torch.empty(log_vec.numel(), dtype=torch.float32).pin_memory() for _ in range(2)],
                               [torch.cuda.Event() for _ in range(2)], 0]
        i = ent[2]
        ent[2] = 1 - i
        host, ev = ent[0][i], ent[1][i]
        host.copy_(log_vec.detach().reshape(-1), non_blocking=True)
        ev.record()

        def read():
            ev.synchronize()
            return host.tolist()
        return read

    def train_batch_async(self, batch_data):
        """train_batch in two halves: the step is QUEUED here (captured steps: no host synchronisation at all) and the returned
        callable hands out its dict of logs, waiting for the device only then.  train() queues step i + 1 before it reads step
        i: the device no longer idles through the host's read-back, bookkeeping and launch of every step (~45 us of a 0.35 ms
        fine-tuning step, a third of a 0.15 ms pointwise step).  Same steps, same numbers, same order of updates; a NaN is
        reported one step later than it happened."""
        args = self.args
        log = {}
        channelled = hasattr(batch_data, 'mask_p_pred') and hasattr(batch_data, 'waypoints')
        from .. import hip_graphs_safe
        if channelled and getattr(args, 'hip_graph', True) and batch_data.position.is_cuda and hip_graphs_safe():
            out, aux = self._graphed_rollout_step(batch_data)
            self._carry_dest_idx(batch_data, aux)
            read = self._log_readback(aux['log_vec'])
            epoch, batch_idx = getattr(self, 'epoch', None), getattr(self, 'batch_idx', None)

            def resolve():
                vals = read()                             # the step's only host synchronisation
                assert not vals[9], f'find nan in epoch : {epoch} {batch_idx}'
                self.collision_count += vals[7]
                self.hard_collision_count += vals[8]
                names = ('loss', 'mse', 'collision', 'hard_collision', 'collision_pred', 'acc_pred', 'reg')
                log.update(dict(zip(names, vals[:7])))
                log['n'] = int(vals[10])
                return log
            return resolve
        self.optimizer.zero_grad()
        if channelled:                                                                     # channelled windows
            # the same gradient bookkeeping as the captured step (the frames' weight gradients summed by the slot-sum launches, those
            # riding in the relfeat backward's launches): eager and captured steps run the same kernels in the same order
            with self._grad_sink.step(), ops.deferred_slot_sums():
                out = self.test_multiple_rollouts_for_training(batch_data)
                out[0].backward()
            names = ('loss', 'mse', 'collision', 'hard_collision', 'collision_pred', 'acc_pred', 'reg')
            log.update({k: float(v.detach()) for k, v in zip(names, out)})
            log['n'] = int(torch.sum(batch_data.mask_p_pred == 1).item())
            self.optimizer.step()
            return lambda: log
        else:                                                                              # pointwise rows
            # (only with the hand-written kernels: capturing the library-GEMM path here segfaulted in hipStreamEndCapture
            # when an earlier capture of the process had used other GEMM selections -- that path stays eager, as before)
            graphed = (getattr(args, 'hip_graph', True) and getattr(args, 'hip_graph_pointwise', True)
                       and batch_data[0].is_cuda and hip_graphs_safe() and self._capturable() and all(self._path_flags()))
            if graphed:
                (loss, mse_loss, reg, cp), log_vec = self._graphed_pointwise_step(tuple(batch_data))
                read = self._log_readback(log_vec)
                rows = int(batch_data[3].shape[0])

                def resolve():
                    vals = read()                          # the step's only host synchronisation
                    if reg is not None:
                        log['reg'] = vals[2]
                    if cp is not None:
                        log['collision_pred'] = vals[3]
                    log.update(loss=vals[0], mse=vals[1], n=rows)
                    return log
                return resolve
            loss, mse_loss, reg, cp = self._pointwise_terms(batch_data)
            if reg is not None:
                log['reg'] = float(reg.detach())
            if cp is not None:
                log['collision_pred'] = float(cp.detach())
            log.update(loss=float(loss.detach()), mse=float(mse_loss.detach()), n=int(batch_data[3].shape[0]))
        loss.backward()
        self.optimizer.step()
        return lambda: log

    def train(self, train_loaders, val_data=None, test_data=None, validate_fn=None):
        """Epoch loop with best-validation model selection and the reference's patience rule (incl. its swapped
        patience / ft_patience, simulators.py:393).  As in the reference (simulators.py:291-393) the weights are
        checkpointed on every validation improvement; in the fine-tune stage the incoming (pre-trained) weights are
        checkpointed and validated first, so a fine-tuned model is only kept when it beats them.  On return the
        model holds the BEST weights of this stage (the reference reloads its checkpoint in every consumer:
        finetune(), test_multiple_rollouts(load_model=True), main.py:166)."""
        args = self.args
        start = time.time()
        best, patience = 1e5, 0
        history = []
        validating = validate_fn is not None or val_data is not None

        def run_validation():
            if validate_fn is not None:
                return validate_fn(self)
            return self.validate(val_data)[0]
        saved = False
        if self.finetune_flag and validating:                        # simulators.py:298-304
            self.epoch = 0
            self.save_model(args, self.finetune_flag)
            saved = True
            best = run_validation()
            self.initial_val_loss = best
            self.initial_val_eval = dict(getattr(self, 'last_eval', {}) or {})
            if test_data:
                self.test_multiple_rollouts(test_data, load_model=False, test_flag=True)
        for epoch in range(args.epochs):
            self.epoch, self.collision_count, self.hard_collision_count = epoch, 0, 0
            self.model.train()
            sums, n, batches = {}, 0, 0
            # one step of lookahead: step i + 1 is queued before step i's scalars are read (train_batch_async)
            # (a train_batch that somebody replaced -- a subclass, a test's hook -- is called as it is: one step at a time)
            plain = type(self).train_batch is _TRAIN_BATCH and 'train_batch' not in vars(self)
            waiting = None
            for batch_idx, batch in enumerate(itertools.chain(train_loaders, (None,))):
                nxt = None
                if batch is not None:
                    self.batch_idx = batch_idx
                    if plain:
                        nxt = self.train_batch_async(batch)
                    else:
                        nxt = (lambda done: (lambda: done))(self.train_batch(batch))
                if waiting is not None:
                    log = waiting()
                    n += log.pop('n')
                    batches += 1
                    for k, v in log.items():
                        sums[k] = sums.get(k, 0.0) + v
                    self.time_iter = time.time() - start
                waiting = nxt
            # per simulated (frame, agent) entry like the reference; its acc_pred is a mean over batches (:367)
            epoch_log = {k: v / (max(batches, 1) if k == 'acc_pred' else max(n, 1)) for k, v in sums.items()}
            history.append(epoch_log)
            print('Epoch {}:'.format(epoch))
            print('Time {:.4f} -- Training loss:{}, mse:{}, coll_pred:{}, acc_pred:{}, coll:{}, hard_coll:{}'.format(
                self.time_iter, epoch_log.get('loss'), epoch_log.get('mse'), epoch_log.get('collision_pred', 0.0),
                epoch_log.get('acc_pred', 0.0), epoch_log.get('collision', 0.0), epoch_log.get('hard_collision', 0.0)))
            if self.finetune_flag:
                print('training collision count hard/soft: {} & {}'.format(self.hard_collision_count,
                                                                           self.collision_count))
                epoch_log['train_collisions'] = (self.hard_collision_count, self.collision_count)
            if not validating:
                continue
            val_loss = run_validation()
            epoch_log['val_loss'] = val_loss
            if isinstance(getattr(self, 'last_eval', None), dict) and not self.last_eval.get('test_flag', True):
                epoch_log['val_eval'] = dict(self.last_eval)
            if test_data:
                self.test_multiple_rollouts(test_data, load_model=False, test_flag=True)
                epoch_log['test_eval'] = dict(self.last_eval)
            if val_loss < best:
                print('!!!!!!!!!! Model Saved at epoch {} !!!!!!!!!!'.format(epoch))
                best, patience = val_loss, 0
                self.save_model(args, self.finetune_flag)
                saved = True
                epoch_log['saved'] = True
            else:
                patience += 1
                if patience > (args.patience if self.finetune_flag else args.ft_patience):
                    break
        self.best_val_loss = best
        if saved:                                                    # leave the best weights of this stage in place
            self.load_model(args, set_model=False, finetune_flag=self.finetune_flag)
        return history

    # ---- rollout evaluation and fine-tuning (simulators.py:395-554) ----
    @staticmethod
    def post_process(data, pred_data, pred_mask_p, mask_p):
        """Agents that have arrived (gone in the rollout, still present in the data) are pinned to
        their final destination (simulators.py:442-463)."""
        dest_idx = (data.dest_num - 1).expand(data.waypoints.shape[:-3] + data.dest_num.shape[-1:])
        dest = _gather_waypoints(data.waypoints, dest_idx).unsqueeze(-3).expand_as(pred_data)
        return torch.where(((mask_p == 1) & (pred_mask_p == 0)).unsqueeze(-1), dest, pred_data)

    def test_multiple_rollouts(self, data, load_model=True, test_flag=True, reduction='sum'):
        """Roll every clip of `data` (a list of TimeIndexedPedData, or one clip: the reference's other branch) from
        `skip_frames` and score it: (loss, mse, mae, ot, mmd) like simulators.py:465-554.  MAE is the mean displacement (the
        reference has no FDE); OT / MMD are the batched restatements in functions/metrics.py."""
        from ..functions import metrics as METRIC
        args = self.args
        self.model.eval()
        if not isinstance(data, list):
            # single-clip branch (simulators.py:471-490): predictions shifted by one frame against the labels,
            # 'mean' reductions over the simulated entries, collisions at a fixed 0.6 m threshold
            with torch.no_grad():
                p_pred = self.get_multiple_rollouts(data, t_start=args.skip_frames, load_model=load_model).position
                p_pred = torch.cat((p_pred[1:], p_pred[-1:]), dim=0)
                mask = data.mask_p_pred.long()
                labels = data.labels[..., :2]
                sel = mask == 1
                loss = mse = F.mse_loss(p_pred[sel], labels[sel], reduction='mean').item()
                mae = METRIC.mae_with_time_mask(p_pred, labels, mask, reduction='mean')
                ot = METRIC.ot_with_time_mask(p_pred, labels, mask, reduction='mean')
                mmd = METRIC.mmd_with_time_mask(p_pred, labels, mask, reduction='mean')
                collision = METRIC.collision_count(p_pred, 0.6, reduction='sum')
            if test_flag:
                print('---------------------------------------')
                print('Test loss:{}, test_mse:{}, test_mae:{}, test ot:{}, test mmd:{}'.format(loss, mse, mae, ot, mmd))
            print('test/val collision count (0.6 m): {}'.format(collision))
            return loss, mse, mae, ot, mmd
        clips = data
        loss_sum = mse_sum = mae_sum = ot_sum = mmd_sum = fde_sum = 0.0
        frames = fde_n = 0
        coll = hard = 0.0
        n = 0
        for d in clips:
            with torch.no_grad():
                pred = self.get_multiple_rollouts(d, t_start=args.skip_frames, load_model=load_model)
                p_pred = pred.position
                mask = d.mask_p_pred.long()
                c = METRIC.collision_count(p_pred[args.skip_frames:], args.collision_threshold, reduction='sum')
                h = METRIC.collision_count(p_pred[args.skip_frames:], args.collision_threshold / 2, reduction='sum')
                coll, hard = coll + c, hard + h
                p_pred = self.post_process(d, p_pred, pred.mask_p, mask)
                labels = d.labels[..., :2]
                m = (mask == 1).unsqueeze(-1)
                mse = torch.where(m, (p_pred - labels) ** 2, torch.zeros_like(p_pred)).sum().item()
                loss = mse + (0 if test_flag else args.val_coll_weight * (c + h))
                if test_flag:
                    fde_sum += METRIC.fde_with_time_mask(p_pred, labels, mask, reduction='sum')
                    fde_n += int(((mask == 1).sum(dim=0) > 0).sum().item())
                    mae_sum += METRIC.mae_with_time_mask(p_pred, labels, mask, reduction='sum')
                    ot_sum += METRIC.ot_with_time_mask(p_pred, labels, mask, reduction='sum')
                    mmd_sum += METRIC.mmd_with_time_mask(p_pred, labels, mask, reduction='sum')
                n += int((mask == 1).sum().item())
                frames += int((mask.sum(dim=1) > 0).sum().item())
                loss_sum, mse_sum = loss_sum + loss, mse_sum + mse
        n, frames = max(n, 1), max(frames, 1)
        loss, mse, mae = loss_sum / n, mse_sum / n, mae_sum / n
        ot, mmd = ot_sum / frames, mmd_sum / frames
        if test_flag:
            print('---------------------------------------')
            print('Test loss:{}, test_mse:{}, test_mae:{}, test ot:{}, test mmd:{}'.format(loss, mse, mae, ot, mmd))
        print('test/val collision count hard/soft: {} & {}'.format(hard, coll))
        # everything of this evaluation in one place (the reference only prints the collision counts); FDE is this
        # repository's addition (BASELINE.json configs[4] asks for ADE/FDE; the reference's "MAE" is the ADE)
        self.last_eval = dict(loss=loss, mse=mse, mae=mae, ot=ot, mmd=mmd, collisions=coll, hard_collisions=hard,
                              fde=(fde_sum / max(fde_n, 1)) if test_flag else None, test_flag=bool(test_flag))
        return loss, mse, mae, ot, mmd

    def validate(self, val_data):
        if isinstance(val_data, list):
            val_loss, val_mse = self.test_multiple_rollouts(val_data, load_model=False, test_flag=False)[:2]
        else:
            val_loss, val_mse = self.test_pointwise(val_data)
        print('Time {:.4f} -- Validation loss:{}, val_mse:{}'.format(self.time_iter, val_loss, val_mse))
        return val_loss, val_mse

    def finetune(self, train_loaders, val_data, test_data, pretrained_state=None):
        """Rollout fine-tuning (simulators.py:395-428): switch to the fine-tune network / optimiser, load the
        intersection of the BEST pre-trained weights (the reference reads its pre-training checkpoint from disk),
        validate them first, train on channelled windows, finish on the best fine-tuned weights and test them."""
        args = self.args
        if pretrained_state is not None:
            pretrained = pretrained_state
        elif self._checkpoints.get(False) is not None:
            pretrained = self._checkpoints[False]
        elif getattr(args, 'save_dir', None) and os.path.exists(self._ckpt_path(args, False)):
            pretrained = torch.load(self._ckpt_path(args, False), map_location=args.device)
        else:
            pretrained = self.model.state_dict()                     # fresh simulator: whatever the model holds
        pretrained = {k: v.detach().clone() for k, v in pretrained.items()}
        self.set_ft_model(args)
        self.set_ft_optimizer(args)
        self.set_ft_scheduler(args)
        own = self.model.state_dict()
        own.update({k: v for k, v in pretrained.items() if k in own})
        self.model.load_state_dict(own)
        self.finetune_flag = True
        history = self.train(train_loaders, val_data if val_data else None, test_data,
                             validate_fn=(lambda s: s.validate(val_data)[0]) if val_data else None)
        result = None
        if test_data:
            result = self.test_multiple_rollouts(test_data, load_model=False)
        self.finetune_flag = False
        self.finetune_test_result = result
        return history


_TRAIN_BATCH = BaseSimulator.train_batch      # (train() pipelines its steps only while this is still the method it calls)

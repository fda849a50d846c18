"""PINNSF network family (reference src/models/model.py:16-119, 720-792, 950-1305).

The modules hold the parameters (nn.Linear, so the reference's state_dicts load unchanged); on the GPU their
arithmetic runs on this package's hand-written matrix-core kernels (piml_amd.ops.fused_pinnsf / fused_encoders /
fused_row_decoder: encoders, processor incl. its train-mode dropout, neighbour-axis sum, decoders, predictors,
collision head, desired force) for the reference's default geometry, and on library GEMMs + HIP glue kernels for
every other geometry.  Class names, constructor `args` fields, forward input/output lists and state_dict keys match
the reference (tests/test_models.py loads reference-initialised state_dicts and reproduces the reference's outputs).

Reference behaviours kept on purpose:
  Q2  the desired-force normalisation reduces over dim=1 (model.py:1290): the per-agent norm
      for (N,7) inputs, a norm over AGENTS for channelled (C,N,7) inputs;
      `fix_dest_norm=True` on the module switches to the per-agent norm.
  Q3  ResDNN feeds every block the ORIGINAL input and keeps only the last block's output
      (model.py:115-119); blocks 1.. have no layers, so with >= 2 "layers" the processor is
      Dropout(2x) and resnet.0's weights are dead (but present in the state_dict).
  Q4  zero-padded neighbour rows still pass through the encoder (bias-driven constants).
Deviation: in train mode the reference calls Dropout once per (discarded) block; here only the
surviving block is evaluated, and on the fused GPU path the keep-mask is drawn by this package's own Philox kernel
(ops.dropout_keep_bits), so the RNG stream differs (eval mode is exact; `ResDNN.keep_bits` injects a given mask).
"""
import contextlib
import logging
import os
import types

_LOG = logging.getLogger('piml_amd')
_NOTED = set()


def _note_fallback(what):
    """Say ONCE per reason that a GPU forward pass left the hand-written kernels for the library-GEMM path."""
    if what not in _NOTED:
        _NOTED.add(what)
        _LOG.warning('piml_amd: %s -- this forward pass runs on library GEMMs + glue kernels, not on the fused '
                     'matrix-core kernels', what)

import torch
import torch.nn as nn

# On the GPU the elementwise / reduction glue around the GEMMs (ReLU backward + bias-gradient sums,
# processor scaling + neighbour-axis sum, desired-force epilogue) runs as fused HIP kernels
# (piml_amd/csrc/mlpglue.hip); the GEMMs remain torch.addmm / torch.mm.  PIML_FUSED_GLUE=0 (or
# setting this flag to False) keeps the plain torch.nn expression of the same arithmetic.
FUSED_GLUE = os.environ.get('PIML_FUSED_GLUE', '1') != '0'
# The encoders themselves (Linear(in, 128) ReLU Linear(128, 128) ReLU Linear(128, 128), the processor's `2 x` and the
# neighbour-axis sum) run as ONE hand-written f32-MFMA kernel per direction for both branches
# (piml_amd/csrc/encoder.hip, ops.fused_encoders) when the network has the reference's default geometry;
# PIML_FUSED_ENCODER=0 keeps the library-GEMM chain.  Below FUSED_ENCODER_MIN_ROWS neighbour rows the launch is
# latency-bound either way and the library chain is kept.
FUSED_ENCODER = os.environ.get('PIML_FUSED_ENCODER', '1') != '0'
FUSED_ENCODER_MIN_ROWS = int(os.environ.get('PIML_FUSED_ENCODER_MIN_ROWS', '512'))
# ... and for `pinnsf` / `pinnsf_m` the decoder tail too (piml_amd/csrc/decoder.hip): the whole network is one autograd
# node (ops.fused_pinnsf).  PIML_FUSED_NETWORK=0 keeps the decoders / predictors on library GEMMs.
FUSED_NETWORK = os.environ.get('PIML_FUSED_NETWORK', '1') != '0'
# bottleneck variants: decoder + predictor per neighbour row on the fused kernels (PIML_FUSED_ROW_DECODER=0: library GEMMs)
FUSED_ROW_DECODER = os.environ.get('PIML_FUSED_ROW_DECODER', '1') != '0'
# ... and their neighbour-axis sums + desired-force term in one launch (PIML_FUSED_KSUM_TAIL=0: torch .sum + the plain epilogue)
FUSED_KSUM_TAIL = os.environ.get('PIML_FUSED_KSUM_TAIL', '1') != '0'
# pinnsf_res's corrector (attention pooling + 128 -> 64 -> 2 tail) on ops.fused_corrector (csrc/corrector.hip)
FUSED_CORRECTOR = os.environ.get('PIML_FUSED_CORRECTOR', '1') != '0'
PREPACK = os.environ.get('PIML_PREPACK', '1') != '0'          # packed_weights(): pack once per block
# inference frames (predictions only, eval): neighbour-axis sum before the encoders' last layer (ops.fused_pinnsf_pooled)
POOLED_INFERENCE = os.environ.get('PIML_POOLED_INFERENCE', '1') != '0'
TAIL_IN_STEP_MAX_AGENTS = 256          # agents per slice up to which a training frame's tail rides in the frame step's launches


def activation_layer(act_name, negative_slope=0.1):
    if isinstance(act_name, str):
        table = {'sigmoid': nn.Sigmoid, 'relu': nn.ReLU}
        if act_name.lower() in table:
            return table[act_name.lower()]()
        if act_name.lower() == 'leaky_relu':
            return nn.LeakyReLU(negative_slope)
        raise NotImplementedError(act_name)
    if isinstance(act_name, type) and issubclass(act_name, nn.Module):
        return act_name()
    raise NotImplementedError(act_name)


class MLP(nn.Module):
    """Linear layers at even indices of `self.mlp`, activations at odd ones (model.py:40-65)."""

    def __init__(self, input_size, layer_sizes, activation=None, dropout=0, output_act=None):
        super().__init__()
        if dropout:
            raise NotImplementedError('MLP dropout > 0 is unreachable in the reference (model.py:60-61 raises)')
        self.dropout = dropout
        sizes = [input_size] + list(layer_sizes)
        mods = []
        for i in range(len(sizes) - 1):
            last = i == len(sizes) - 2
            act = (output_act if output_act is not None else nn.Identity()) if last else \
                (activation if activation is not None else nn.ReLU())
            mods += [nn.Linear(sizes[i], sizes[i + 1]), act]
        self.mlp = nn.Sequential(*mods)
        self._fusable = all(isinstance(a, (nn.ReLU, nn.Identity)) for a in mods[1::2])

    def fused_ok(self, x):
        return FUSED_GLUE and self._fusable and x.is_cuda and x.dtype == torch.float32

    def forward(self, x, defer_last_bias=False):
        """defer_last_bias (fused GPU path only, last layer without activation): return the output WITHOUT the
        last layer's bias -- the caller adds `self.mlp[-2].bias` (ops.scale_ksum does, in its own pass)."""
        if self.fused_ok(x):
            from .. import ops
            lins = self.mlp[0::2]
            relus = [isinstance(a, nn.ReLU) for a in self.mlp[1::2]]
            return ops.mlp_chain(x, relus, *[t for lin in lins for t in (lin.weight, lin.bias)],
                                 defer_last_bias=defer_last_bias)
        assert not defer_last_bias
        return self.mlp(x)


class ResBlock(nn.Module):
    def __init__(self, in_dim, hidden_units, activation, dropout=0, use_bn=False):
        super().__init__()
        if use_bn:
            raise NotImplementedError('bn in resblock has not been implemented!')
        self.lin = MLP(in_dim, hidden_units, activation, dropout, activation)

    def forward(self, x):
        return self.lin(x) + x


class ResDNN(nn.Module):
    """model.py:82-119, including quirk Q3 (see module docstring)."""

    def __init__(self, input_dim, hidden_units, activation=None, dropout=0, use_bn=False):
        super().__init__()
        hidden_units = [list(h) for h in hidden_units]
        if input_dim != hidden_units[0][0]:
            raise ValueError('In ResBlock, the feature size must be equal to the hidden size! '
                             'input_dim:{}, hidden_size: {}'.format(input_dim, hidden_units[0]))
        activation = activation if activation is not None else nn.ReLU()
        self.dropout = nn.Dropout(dropout)
        hidden_units[0] = [input_dim] + hidden_units[0]
        self.hidden_units = hidden_units
        # the reference passes `use_bn` in ResBlock's dropout slot (model.py:113)
        self.resnet = nn.ModuleList([ResBlock(h[0], h[1:], activation, use_bn) for h in hidden_units])

    # tests / hosts that draw their own mask: int32 bits (rows, ceil(width / 32)) in ops.dropout_keep_bits' layout, used
    # by the next train-mode forward passes instead of a fresh draw (both on the fused kernels and in `forward` below)
    keep_bits = None

    def forward(self, x):
        out = self.resnet[len(self.hidden_units) - 1](x)
        if self.keep_bits is not None and self.training and 0 < self.dropout.p:
            from .. import ops
            keep = ops.unpack_keep_bits(self.keep_bits, out.shape[-1]).view(out.shape)
            return out * keep / (1.0 - self.dropout.p) if self.dropout.p < 1 else out * 0.0
        return self.dropout(out)

    def scales_input(self):
        """True when this module computes keep * scale * x (quirk Q3: >= 2 "layers" = Dropout(2 x))."""
        return len(self.hidden_units) >= 2

    def dropout_active(self):
        return self.training and self.dropout.p > 0

    def fused_spec(self, rows, device, in_launch=False, stream_id=0):
        """(scale, keep) of `keep * scale * x` for a (rows, width) input on `device`, or None when this module is not of
        that form.  keep is None without active dropout (eval mode or p = 0: scale = 2).  In train mode scale = 2 / (1 - p)
        and keep is the injected `self.keep_bits`, or a fresh draw: bits from ops.dropout_keep_bits (one small launch), or
        -- in_launch=True, for ops.fused_encoders / fused_pinnsf -- the request ('draw', p) that makes the encoder launch
        draw the mask itself."""
        if not self.scales_input():
            return None
        if not self.dropout_active():
            return 2.0, None
        p = float(self.dropout.p)
        scale = 2.0 / (1.0 - p) if p < 1 else 0.0
        if self.keep_bits is not None:
            return scale, self.keep_bits
        if in_launch:
            return scale, ('draw', p)
        from .. import ops
        return scale, ops.dropout_keep_bits(rows, self.hidden_units[-1][-1], p, device, stream_id)


class attn_pooling(nn.Module):
    """model.py:950-970: softmax(exp(w(x))) pooling over the neighbour axis."""

    def __init__(self, dim):
        super().__init__()
        self.get_weights = MLP(dim, [dim, 1])

    def forward(self, x):
        attn = torch.softmax(torch.exp(self.get_weights(x)), dim=-2)
        return torch.matmul(x.transpose(-1, -2), attn).squeeze()


class _PINNSFBase(nn.Module):
    """Shared body of the PINNSF variants; subclasses set the class attributes below."""
    bottleneck = False          # decoder + predictor applied per neighbour, then summed over k
    collision_head = None       # None | 'msgs' (pinnsf_m) | 'decoded' (pinnsf_bm)
    defer_ksum_epilogue = False # set around an inference frame's forward by BaseSimulator: the bottleneck variants leave their epilogue
    pending_ksum = None         # (neighbour-axis sums + desired force) to the integrator launch and park its operands here
    defer_train_tail = False    # set around a TRAINING rollout frame's forward by BaseSimulator (fused frame step): under the reference's
    pending_tail = None         # agent-axis norm (channelled input, quirk Q2) the tail -- neighbour-axis sums + desired force -- is left
                                # to the frame step's launch (ops.rollout_frame tail=): operands parked here, out[0] is None
    _ph2 = None                 # folded weights + operand images of the pooled inference path (see _pooled_inference)
    predictions_only = False    # set by the inference rollouts (BaseSimulator): the auxiliary collision head (`pinnsf_m`: on the messages,
                                # `pinnsf_bm`: on the decoded rows) is not launched -- those loops read out[0] only (simulators.py:602) --
                                # and forward returns None in its place
    messages_wanted = True      # False: the caller reads predictions[0] (and the collision head's output, without training it) only --
                                # what the reference's loops do unless reg_weight > 0 (src/models/simulators.py:331-347, :702-737).
                                # `pinnsf` / `pinnsf_m` then run on the agents' SUMS of h2 where no dropout mask is active and the
                                # library serves the shape (ops.fused_pinnsf(sums=True), PIML_POOL_TRAIN), and under a dropout mask on
                                # the sums of the messages the encoder forward leaves (PIML_POOL_MSGS); out[1] / out[2] are None
    residual = False            # pinnsf_res corrector branch
    obs_encoder_in = 6          # PINNSF_residual uses args.obs_feature_dim instead
    taus = (2, 2)               # (non-ucy tau, ucy tau)

    def __init__(self, args):
        super().__init__()
        ucy = getattr(args, 'dataset_name', None) in {'ucy'}
        self.tau = self.taus[1] if ucy else self.taus[0]
        self.fix_dest_norm = False
        self._ones_bits = None      # pinnsf_res: cached all-ones keep-mask of the pedestrian branch (see forward)
        # optional: a side stream on which the obstacle branch runs concurrently with the pedestrian
        # branch (the two are independent until their accelerations are added; inside a captured
        # HIP graph this becomes two parallel chains of GEMMs that fill the 256 CUs better)
        self.obs_stream = None
        self._packs = None          # ops.PinnsfPacks of `packed_weights()`
        self.ped_feature_dim = args.ped_feature_dim
        self.obs_feature_dim = args.obs_feature_dim
        self.self_feature_dim = args.self_feature_dim
        enc = [args.encoder_hidden_size] * args.encoder_hidden_layers
        pro = [[args.processor_hidden_size] for _ in range(args.processor_hidden_layers)]
        dec = [args.decoder_hidden_size] * args.decoder_hidden_layers
        act = activation_layer(args.activation)

        self.ped_encoder = MLP(self.ped_feature_dim, enc)
        obs_in = self.obs_encoder_in if self.obs_encoder_in else self.obs_feature_dim
        if self.obs_encoder_in or self.obs_feature_dim > 0:
            self.obs_encoder = MLP(obs_in, enc)
        self.ped_processor = ResDNN(enc[-1], pro, act, args.dropout)
        self.obs_processor = ResDNN(enc[-1], [list(h) for h in pro], act, args.dropout)
        self.ped_decoder = MLP(pro[-1][-1], dec)
        self.obs_decoder = MLP(pro[-1][-1], dec)
        self.ped_predictor = MLP(dec[-1], [2])
        self.obs_predictor = MLP(dec[-1], [2])
        if self.collision_head == 'msgs':
            self.ped_collision_predictor = MLP(pro[-1][-1], [dec[-1], 1])
        elif self.collision_head == 'decoded':
            self.ped_collision_predictor = MLP(dec[-1], [dec[-1], 1])
        if self.residual:
            res = [[args.processor_hidden_size] for _ in range(args.res_hidden_layers)]
            self.corrector = nn.ModuleList([
                ResDNN(enc[-1], res, act, args.dropout), attn_pooling(res[-1][-1]),
                MLP(res[-1][-1], [int(res[-1][-1] / 2), 2])])

    def desired_force(self, self_features):
        """(v0 * e - v) / tau with e = dest_features / |dest_features| (model.py:1289-1294)."""
        desired_speed = self_features[..., -1].unsqueeze(-1)
        dim = -1 if self.fix_dest_norm else 1                       # quirk Q2
        temp = torch.norm(self_features[..., :2], p=2, dim=dim, keepdim=True)
        temp = torch.where(temp == 0, temp + 0.1, temp)
        dest_direction = self_features[..., :2] / temp
        return (desired_speed * dest_direction - self_features[..., 2:4]) / self.tau

    @staticmethod
    def _process_and_pool(processor, encoded, bias=None):
        """(processor(encoded [+ bias]), its sum over the neighbour axis)."""
        if FUSED_GLUE and processor.scales_input() and encoded.is_cuda and encoded.shape[-1] % 4 == 0 \
                and encoded.dtype == torch.float32:
            from .. import ops
            scale, keep = processor.fused_spec(encoded.numel() // encoded.shape[-1], encoded.device)
            return ops.scale_ksum(encoded, scale, bias=bias, keep_bits=keep)
        assert bias is None
        emb = processor(encoded)
        return emb, emb.sum(dim=-2)

    def _encode_process_pool(self, feats, encoder, processor):
        """encoder -> processor -> k-sum.  On the fused GPU path the encoder's last Linear runs as a plain GEMM
        (its bias-epilogue variant is the slower library kernel) and its bias is added inside the k-sum pass."""
        last = encoder.mlp[-2] if len(encoder.mlp) >= 2 else None
        if last is not None and encoder.fused_ok(feats) and isinstance(encoder.mlp[-1], nn.Identity) \
                and processor.scales_input() and last.out_features % 4 == 0:
            if 256 % (last.out_features // 4) == 0 and feats.dim() >= 3:      # one autograd node for all of it
                from .. import ops
                lins = encoder.mlp[0::2]
                relus = [isinstance(a, nn.ReLU) for a in encoder.mlp[1::2]]
                scale, keep = processor.fused_spec(feats.numel() // feats.shape[-1], feats.device)
                return ops.encoder_pool(feats, relus, scale, *[t for lin in lins for t in (lin.weight, lin.bias)],
                                        keep_bits=keep)
            return self._process_and_pool(processor, encoder(feats, defer_last_bias=True), bias=last.bias)
        return self._process_and_pool(processor, encoder(feats))

    @staticmethod
    def _encoder_fusable(feats, encoder, processor):
        """The fused f32-MFMA encoder kernel covers exactly the reference's default encoder geometry."""
        if not (FUSED_GLUE and FUSED_ENCODER and feats.is_cuda and feats.dtype == torch.float32 and feats.dim() >= 3):
            return False
        lins, acts = encoder.mlp[0::2], encoder.mlp[1::2]
        return (len(lins) == 3 and 1 <= feats.shape[-1] <= 8 and lins[0].in_features == feats.shape[-1]
                and all(lin.out_features == 128 for lin in lins) and isinstance(acts[0], nn.ReLU)
                and isinstance(acts[1], nn.ReLU) and isinstance(acts[2], nn.Identity)
                and processor.scales_input() and processor.hidden_units[-1][-1] == 128
                and feats.numel() // feats.shape[-1] >= FUSED_ENCODER_MIN_ROWS)

    def _fused_encoders(self, ped_features, obs_features):
        """{'ped': (msgs, pooled), 'obs': (...)} for the branches the fused kernel takes (both in one launch)."""
        cand = [('ped', ped_features, self.ped_encoder, self.ped_processor)]
        if self.obs_feature_dim > 0:
            cand.append(('obs', obs_features, self.obs_encoder, self.obs_processor))
        use = [c for c in cand if self._encoder_fusable(*c[1:])]
        if len({p.dropout_active() for _, _, _, p in use}) > 1:       # one launch = dropout on every branch or on none
            use = use[:1]
        if not use:
            return {}
        from .. import ops
        specs = self._launch_specs([(p, f) for _, f, _, p in use])
        packs = self._active_packs() if [c[0] for c in use] == [c[0] for c in cand] else None     # pack order = branch order
        res = ops.fused_encoders([dict(x=f, scale=sp[0], keep_bits=sp[1], pooled=not self.bottleneck,
                                       weights=[t for lin in e.mlp[0::2] for t in (lin.weight, lin.bias)])
                                  for (_, f, e, p), sp in zip(use, specs)], packs=packs)
        return {c[0]: r for c, r in zip(use, res)}

    @staticmethod
    def _launch_specs(pairs):
        """fused_spec of every (processor, features) branch of ONE encoder launch: the launch draws the masks itself unless
        any branch carries injected bits (then the others get theirs from the stand-alone generator: one kind per launch)."""
        injected = any(p.keep_bits is not None for p, _ in pairs)
        return [p.fused_spec(f.numel() // f.shape[-1], f.device, in_launch=not injected, stream_id=i)
                for i, (p, f) in enumerate(pairs)]

    def _fused_row_decoders(self, pre):
        """Bottleneck variants: decoder + predictor per neighbour row on the fused kernels (ops.fused_row_decoder), for the
        branches whose embeddings the fused encoder produced -- both in one launch.  {} when not applicable."""
        if not (self.bottleneck and pre and FUSED_ROW_DECODER):
            return {}
        names, brs = [], []
        for name, d, q in (('ped', self.ped_decoder, self.ped_predictor),
                           ('obs', getattr(self, 'obs_decoder', None), getattr(self, 'obs_predictor', None))):
            if name not in pre or d is None:
                continue
            emb = pre[name][0]
            dl, da = d.mlp[0::2], d.mlp[1::2]
            ok = (emb.is_cuda and emb.dtype == torch.float32 and emb.shape[-1] == 128 and len(dl) == 2
                  and (dl[0].in_features, dl[0].out_features, dl[1].in_features, dl[1].out_features) == (128, 64, 64, 64)
                  and isinstance(da[0], nn.ReLU) and isinstance(da[1], nn.Identity) and len(q.mlp) == 2
                  and (q.mlp[0].in_features, q.mlp[0].out_features) == (64, 2) and isinstance(q.mlp[1], nn.Identity)
                  and emb.numel() // 128 >= FUSED_ENCODER_MIN_ROWS)
            if ok:
                names.append(name)
                brs.append(dict(emb=emb, decoder=[t for lin in dl for t in (lin.weight, lin.bias)],
                                predictor=[q.mlp[0].weight, q.mlp[0].bias]))
        if not brs:
            return {}
        from .. import ops
        full = ['ped'] + (['obs'] if self.obs_feature_dim > 0 else [])
        return dict(zip(names, ops.fused_row_decoder(brs, packs=self._active_packs() if names == full else None)))

    def _branch(self, feats, encoder, processor, decoder, predictor, pre=None, rowdec=None, want_sum=True):
        """`pre` = (processor(encoder(feats)), its neighbour-axis sum) when the fused encoder kernel produced them;
        `rowdec` = (predictor(decoder(emb)), decoder(emb)) when the fused row decoder did; want_sum=False: the caller
        sums the per-neighbour outputs itself (ops.pinnsf_epilogue_ksum)."""
        if self.bottleneck:
            emb = pre[0] if pre is not None else processor(encoder(feats))
            if rowdec is not None:
                msgs, decoded = rowdec
            else:
                decoded = decoder(emb)
                msgs = predictor(decoded)
            return (msgs.sum(dim=-2) if want_sum else None), msgs, decoded, emb
        emb, pooled = pre if pre is not None else self._encode_process_pool(feats, encoder, processor)
        return self._decode_pooled(pooled, decoder, predictor), emb, None, emb

    def _decode_pooled(self, pooled, decoder, predictor):
        """predictor(decoder(pooled)) of a non-bottleneck branch outside the whole-network operator (`pinnsf_res`): the row
        decoder kernels with the agents in the role of the rows (ops.fused_row_decoder) when the geometry is the reference's
        128 -> 64 -> 64 -> 2, else the library layers."""
        dl, da = decoder.mlp[0::2], decoder.mlp[1::2]
        if (FUSED_ROW_DECODER and FUSED_GLUE and pooled.is_cuda and pooled.dtype == torch.float32 and pooled.shape[-1] == 128
                and len(dl) == 2 and (dl[0].in_features, dl[0].out_features, dl[1].in_features, dl[1].out_features) == (128, 64, 64, 64)
                and isinstance(da[0], nn.ReLU) and isinstance(da[1], nn.Identity) and len(predictor.mlp) == 2
                and (predictor.mlp[0].in_features, predictor.mlp[0].out_features) == (64, 2)
                and isinstance(predictor.mlp[1], nn.Identity) and pooled.numel() // 128 >= FUSED_ENCODER_MIN_ROWS):
            from .. import ops
            return ops.fused_row_decoder([dict(emb=pooled, decoder=[t for lin in dl for t in (lin.weight, lin.bias)],
                                               predictor=[predictor.mlp[0].weight, predictor.mlp[0].bias])])[0][0]
        return predictor(decoder(pooled))

    def _fused_network(self, ped_features, obs_features, self_features):
        """`pinnsf` / `pinnsf_m` with the reference's default geometry: the whole network -- both encoders, processor
        scale, neighbour-axis sum, decoders, predictors and the desired-force term -- as one autograd node on the fused
        f32-MFMA kernels (ops.fused_pinnsf).  Returns None when the configuration is not covered."""
        if self.bottleneck or self.residual or not FUSED_NETWORK:
            return None
        cand = [(ped_features, self.ped_encoder, self.ped_processor, self.ped_decoder, self.ped_predictor)]
        if self.obs_feature_dim > 0:
            cand.append((obs_features, self.obs_encoder, self.obs_processor, self.obs_decoder, self.obs_predictor))
        if not (self_features.is_cuda and self_features.dtype == torch.float32):
            return None
        for f, e, p, d, q in cand:
            dl, da = d.mlp[0::2], d.mlp[1::2]
            if not (self._encoder_fusable(f, e, p) and tuple(f.shape[:-2]) == tuple(self_features.shape[:-1])
                    and len(dl) == 2 and (dl[0].in_features, dl[0].out_features, dl[1].out_features) == (128, 64, 64)
                    and isinstance(da[0], nn.ReLU) and isinstance(da[1], nn.Identity) and len(q.mlp) == 2
                    and (q.mlp[0].in_features, q.mlp[0].out_features) == (64, 2) and isinstance(q.mlp[1], nn.Identity)):
                return None
        if len({p.dropout_active() for _, _, p, _, _ in cand}) > 1:
            return None
        from .. import ops
        fold = self_features.dim() == 2 or self.fix_dest_norm          # per-row |dest|; else quirk Q2 below
        head = None if self.predictions_only else self._fusable_head()
        packs = self._active_packs()
        specs = self._launch_specs([(p, f) for f, _, p, _, _ in cand])
        if self.predictions_only and POOLED_INFERENCE and packs is not None and not torch.is_grad_enabled() \
                and all(sp[1] is None for sp in specs):
            acc = self._pooled_inference(cand, specs, self_features, fold)
            if acc is not None:
                if not fold:
                    if self_features.dim() == 3:
                        acc = ops.pinnsf_epilogue(acc, None, self_features, self.tau, agent_norm=True)
                    else:
                        acc = acc + self.desired_force(self_features)
                return [acc] + [None] * (len(cand) + (1 if self.collision_head is not None else 0))
        res = ops.fused_pinnsf(
            [dict(x=f, scale=sp[0], keep_bits=sp[1], encoder=[t for lin in e.mlp[0::2] for t in (lin.weight, lin.bias)],
                  decoder=[t for lin in d.mlp[0::2] for t in (lin.weight, lin.bias)],
                  predictor=[q.mlp[0].weight, q.mlp[0].bias]) for (f, e, p, d, q), sp in zip(cand, specs)],
            self_features, self.tau, fold_epilogue=fold, head=head, packs=packs,
            # (a collision head outside the fused geometry reads the messages through torch below: they have to exist)
            sums=not self.messages_wanted and (self.collision_head is None or head is not None or self.predictions_only))
        acc, msgs = res[0], res[1]
        if not fold:
            if self_features.dim() == 3 and self._park_tail(acc, None, self_features):
                acc = None
            elif self_features.dim() == 3:
                acc = ops.pinnsf_epilogue(acc, None, self_features, self.tau, agent_norm=True)
            else:
                acc = acc + self.desired_force(self_features)
        out = [acc, msgs[0]]
        if len(msgs) > 1:
            out.append(msgs[1])
        if self.collision_head is not None:        # 'msgs' (pinnsf_m): head on the pedestrian messages
            if self.predictions_only:              # inference rollouts read the accelerations only: no head workgroups
                out.append(None)
            elif head is not None:
                out.append(res[2].squeeze())
            else:
                out.append(torch.sigmoid(self.ped_collision_predictor(msgs[0])).squeeze())
        return out

    def _park_tail(self, acc_ped, acc_obs, self_features):
        """Leave the agent-norm tail to the caller's frame step?  (training rollout, channelled input, no corrector branch)"""
        # (one workgroup per slice does the tail and the step: ahead of the separate launches while a slice's agents are ONE pass of its
        # 256 threads -- the real clips' 122 agents: 0.346 against 0.364 ms per fine-tuning step -- and behind them at 976 agents, 0.518
        # against 0.498, where the separate step spreads over many workgroups)
        if not (self.defer_train_tail and torch.is_grad_enabled() and self_features.dim() == 3 and not self.fix_dest_norm
                and not self.residual and self_features.is_cuda and self_features.dtype == torch.float32
                and self_features.shape[1] <= TAIL_IN_STEP_MAX_AGENTS):
            return False
        self.pending_tail = (acc_ped, acc_obs, self_features, self.tau)
        return True

    def _pooled_inference(self, cand, specs, self_features, fold):
        """Inference frames (predictions only, eval mode, inside packed_weights()): the neighbour-axis sum BEFORE the
        encoders' last layer, that layer folded into the decoders' first (ops.fused_pinnsf_pooled / PIML_POOL_H2).  The
        folded weights and their operand images are made at the first such frame of a packed_weights() block (outside any
        graph capture) and dropped when the block ends.  None: not served (the message path runs)."""
        from .. import ops
        key = (tuple(f.shape[-2] for f, *_ in cand), tuple(float(sp[0]) for sp in specs))
        st = self._ph2
        if st is None or st.key != key:
            if torch.cuda.is_current_stream_capturing():
                return None
            st = types.SimpleNamespace(key=key, packs=ops.PinnsfPacks(), enc_w=[], dec_w=[])
            for (f, e, p, d, q), sp in zip(cand, specs):
                ew = [t for lin in e.mlp[0::2] for t in (lin.weight, lin.bias)]
                dw = [t for lin in d.mlp[0::2] for t in (lin.weight, lin.bias)] + [q.mlp[0].weight, q.mlp[0].bias]
                st.enc_w.append(ew)
                st.dec_w.append(ops.pooled_h2_decoder_weights(ew, dw, sp[0], f.shape[-2]))
            ops.pinnsf_prepack(st.packs, st.enc_w, st.dec_w, None, defer=False)
            self._ph2 = st
        return ops.fused_pinnsf_pooled(
            [dict(x=f, encoder=ew, decoder=dw) for (f, *_), ew, dw in zip(cand, st.enc_w, st.dec_w)],
            self_features, self.tau, fold_epilogue=fold, packs=st.packs)

    def _correct(self, encoded):
        """corrector[2](corrector[1](corrector[0](encoded)))  (model.py:1050-1052) -- on the hand-written kernels
        (ops.fused_corrector) when the three modules have the reference geometry, else module by module."""
        res, pool, tail = self.corrector[0], self.corrector[1], self.corrector[2]
        gw, tl = pool.get_weights.mlp, tail.mlp
        if FUSED_GLUE and FUSED_CORRECTOR and encoded.is_cuda and encoded.dtype == torch.float32 and encoded.dim() >= 3 \
                and encoded.shape[-1] == 128 and encoded.shape[-2] <= 64 and encoded.numel() > 0 and res.scales_input() \
                and len(gw) == 4 and len(tl) == 4 and isinstance(gw[1], nn.ReLU) and isinstance(gw[3], nn.Identity) \
                and isinstance(tl[1], nn.ReLU) and isinstance(tl[3], nn.Identity) \
                and (gw[0].in_features, gw[0].out_features, gw[2].out_features) == (128, 128, 1) \
                and (tl[0].in_features, tl[0].out_features, tl[2].out_features) == (128, 64, 2) \
                and (not res.dropout_active() or res.dropout.p < 1):
            from .. import ops
            scale, keep = res.fused_spec(encoded.numel() // 128, encoded.device)
            out = ops.fused_corrector(encoded, scale, keep, (gw[0].weight, gw[0].bias, gw[2].weight, gw[2].bias),
                                      (tl[0].weight, tl[0].bias, tl[2].weight, tl[2].bias))
            return out
        return tail(pool(res(encoded)))

    def _fusable_head(self):
        """(w1, b1, w2, b2) of the `pinnsf_m` collision head when it has the reference geometry MLP(128, [64, 1])."""
        if self.collision_head != 'msgs':
            return None
        head = self.ped_collision_predictor.mlp
        if len(head) == 4 and (head[0].in_features, head[0].out_features, head[2].out_features) == (128, 64, 1) \
                and isinstance(head[1], nn.ReLU) and isinstance(head[3], nn.Identity):
            return (head[0].weight, head[0].bias, head[2].weight, head[2].bias)
        return None

    def _pack_spec(self):
        """Weights of the fused kernels in ops.pinnsf_prepack order (pedestrian branch first), or None when the forward
        pass would not use them: the whole network (`_fused_network`), or -- bottleneck variants -- the fused encoders +
        row decoders."""
        if self.residual or not FUSED_ENCODER or not FUSED_GLUE:
            return None
        if self.bottleneck and not FUSED_ROW_DECODER:
            return None
        if not self.bottleneck and not FUSED_NETWORK:
            return None
        if type(self).forward is not _PINNSFBase.forward:            # the polar variants post-process per row
            return None
        cand = [(self.ped_encoder, self.ped_decoder, self.ped_predictor)]
        if self.obs_feature_dim > 0:
            cand.append((self.obs_encoder, self.obs_decoder, self.obs_predictor))
        enc_w, dec_w = [], []
        for e, d, q in cand:
            el, dl = e.mlp[0::2], d.mlp[0::2]
            if len(el) != 3 or len(dl) != 2 or len(q.mlp) != 2 or not el[0].weight.is_cuda \
                    or el[0].weight.dtype != torch.float32 or el[0].in_features > 8:
                return None
            enc_w.append([t for lin in el for t in (lin.weight, lin.bias)])
            dec_w.append([t for lin in dl for t in (lin.weight, lin.bias)] + [q.mlp[0].weight, q.mlp[0].bias])
        return enc_w, dec_w, (None if (self.bottleneck or self.predictions_only) else self._fusable_head())

    def _fold_spec(self):
        """per branch the processor scale when a forward pass inside packed_weights() may run on the agents' sums of h2
        (messages not wanted, no active dropout: the pack then also makes the folded images), else None"""
        if self.messages_wanted or self.bottleneck or self.residual:
            return None
        procs = [self.ped_processor] + ([self.obs_processor] if self.obs_feature_dim > 0 else [])
        if any(p.dropout_active() or not p.scales_input() for p in procs):
            return None
        return [2.0 for _ in procs]

    def _active_packs(self):
        return self._packs if (self._packs is not None and self._packs.active) else None

    @contextlib.contextmanager
    def packed_weights(self):
        """Pack the weights into MFMA operand fragments ONCE for every forward pass inside the block.  For code whose
        weights do not change inside the block: a rollout (no pack launch per frame), or the frames of one
        back-propagated training step (the optimizer step comes after the block).  Without it every forward pass packs
        for itself.  No-op for configurations the fused network does not cover; re-entrant."""
        spec = self._pack_spec()
        if spec is None or not PREPACK or (self._packs is not None and self._packs.active):
            yield
            return
        from .. import ops
        if self._packs is None:
            self._packs = ops.PinnsfPacks()
        ops.pinnsf_prepack(self._packs, *spec, fold=self._fold_spec())
        self._packs.active = True
        try:
            yield
        finally:
            self._packs.active = False
            self._ph2 = None            # the folded inference weights belong to this block's weights
            if getattr(self._packs, 'pending_structs', None) is not None:
                # a deferred pack no forward pass took (PIML_DEFER_PACK): run it now rather than leave raw weight pointers
                # with the library beyond the block (no-op when it has run)
                # (the library keeps one pending pack per DEVICE: flush on the model's, which need not be the current one, and
                # drop the structs -- the only references to the arrays the pending pack points into -- only behind that flush)
                from .. import _lib
                import torch
                with torch.cuda.device(next(self.parameters()).device):
                    _lib.check(_lib.lib().piml_pinnsf_pack_flush(), 'piml_pinnsf_pack_flush')
                self._packs.pending_structs = None

    def forward(self, ped_features, obs_features, self_features):
        assert (self_features.shape[-1] == 7), 'Error: PINN model do not accept inputs of historical velocity'
        fused = self._fused_network(ped_features, obs_features, self_features)
        if fused is not None:
            return fused
        encoded = None
        if self.residual:
            # pinnsf_res (model.py:1024-1059): the corrector reads the pedestrian encoder's RAW output, so that encoder runs
            # on the fused kernels with scale 1 and no mask (its own launch) and the processor is the glue pass
            # (ops.scale_ksum, mask-aware); the obstacle branch takes the standard fused path
            pre = {}
            ped_ok = self._encoder_fusable(ped_features, self.ped_encoder, self.ped_processor)
            obs_ok = self.obs_feature_dim > 0 and self._encoder_fusable(obs_features, self.obs_encoder, self.obs_processor)
            ped_branch = dict(x=ped_features, scale=1.0, pooled=False,
                              weights=[t for lin in self.ped_encoder.mlp[0::2] for t in (lin.weight, lin.bias)])
            spec = None
            if ped_ok and obs_ok and tuple(ped_features.shape[:-2]) == tuple(obs_features.shape[:-2]):
                # BOTH encoders in one launch per direction -- the row count of the many-rows kernels and of the one-pass
                # backward instead of two few-rows launches.  A launch applies one KIND of mask to all its
                # branches: in train mode the obstacle branch's bits are drawn up front (ops.dropout_keep_bits, one small
                # launch) and the pedestrian branch, whose raw output the corrector reads, gets an all-ones mask
                from .. import ops
                rows_o = obs_features.numel() // obs_features.shape[-1]
                scale_o, keep_o = self.obs_processor.fused_spec(rows_o, obs_features.device)
                if keep_o is not None:
                    rows_p = ped_features.numel() // ped_features.shape[-1]
                    key = (rows_p, ped_features.device)
                    if self._ones_bits is None or self._ones_bits[0] != key:
                        self._ones_bits = (key, torch.full((rows_p, 4), -1, dtype=torch.int32, device=ped_features.device))
                    ped_branch['keep_bits'] = self._ones_bits[1]
                both = ops.fused_encoders([ped_branch, dict(x=obs_features, scale=scale_o, keep_bits=keep_o, pooled=True,
                                                            weights=[t for lin in self.obs_encoder.mlp[0::2] for t in (lin.weight, lin.bias)])])
                encoded, pre['obs'] = both[0][0], both[1]
                ped_ok = obs_ok = False
            elif obs_ok:
                spec = self._launch_specs([(self.obs_processor, obs_features)])[0]
            if ped_ok:
                from .. import ops
                encoded = ops.fused_encoders([ped_branch])[0][0]
            if obs_ok:
                from .. import ops
                pre['obs'] = ops.fused_encoders([dict(x=obs_features, scale=spec[0], keep_bits=spec[1], pooled=True,
                                                      weights=[t for lin in self.obs_encoder.mlp[0::2] for t in (lin.weight, lin.bias)])])[0]
        else:
            pre = self._fused_encoders(ped_features, obs_features)
        rowdec = self._fused_row_decoders(pre)
        if FUSED_GLUE and FUSED_ENCODER and ped_features.is_cuda and not pre and encoded is None and \
                ped_features.numel() // ped_features.shape[-1] >= FUSED_ENCODER_MIN_ROWS:
            _note_fallback(f'{type(self).__name__}: geometry / dtype / flags outside the fused encoder kernels '
                           f'(encoder {[lin.out_features for lin in self.ped_encoder.mlp[0::2]]}, '
                           f'{len(self.ped_processor.hidden_units)} processor layers, {ped_features.dtype})')
        # the side stream only pays for the library-GEMM chain; the fused encoder launch already fills the chip
        # (and never while a processor DRAWS a dropout mask: the draw counter is one per device -- offset + ticket,
        # philox.hpp -- and assumes one drawing launch in flight; two branches drawing on two streams would race on it)
        drawing = any(p.dropout_active() and p.keep_bits is None for p in (self.ped_processor, self.obs_processor))
        side = self.obs_stream if (self.obs_feature_dim > 0 and obs_features.is_cuda and not pre and not drawing) else None
        # bottleneck variants with a per-row |dest|: neighbour-axis sums + desired force in one launch (ops.pinnsf_epilogue_ksum)
        # (round 4: also for channelled (C, N, 7) input with the reference's dim=1 norm, quirk Q2 -- the frames of the training rollout)
        ksum_tail = (self.bottleneck and FUSED_GLUE and FUSED_KSUM_TAIL and not self.residual and side is None and self_features.is_cuda
                     and self_features.dtype == torch.float32 and (self_features.dim() in (2, 3) or self.fix_dest_norm))
        acc_o = None
        if self.obs_feature_dim > 0 and side is not None:      # fork: obstacle branch on the side stream
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                acc_o, out_obs_side, _, _ = self._branch(obs_features, self.obs_encoder, self.obs_processor,
                                                         self.obs_decoder, self.obs_predictor)
        if self.residual and encoded is None:
            encoded = self.ped_encoder(ped_features)
        if self.residual:
            emb, pooled = self._process_and_pool(self.ped_processor, encoded)
            ped_msgs = emb
            acc = self._decode_pooled(pooled, self.ped_decoder, self.ped_predictor)
            decoded = None
        else:
            acc, ped_msgs, decoded, emb = self._branch(ped_features, self.ped_encoder, self.ped_processor,
                                                       self.ped_decoder, self.ped_predictor, pre=pre.get('ped'),
                                                       rowdec=rowdec.get('ped'), want_sum=not ksum_tail)
        out_obs = None
        if self.obs_feature_dim > 0:
            if side is not None:                                 # join
                torch.cuda.current_stream().wait_stream(side)
                out_obs = out_obs_side
            else:
                acc_o, out_obs, _, _ = self._branch(obs_features, self.obs_encoder, self.obs_processor,
                                                    self.obs_decoder, self.obs_predictor, pre=pre.get('obs'),
                                                    rowdec=rowdec.get('obs'), want_sum=not ksum_tail)
        if ksum_tail and self.defer_ksum_epilogue and self_features.dim() == 2 and not torch.is_grad_enabled():
            # inference frames: the caller's integrator launch sums the per-neighbour predictions and adds the desired-force term
            # itself (ops.rollout_step ksum=); out[0] is None
            self.pending_ksum = (ped_msgs, out_obs, self.tau)
            predictions = None
        elif ksum_tail and self._park_tail(ped_msgs, out_obs, self_features):
            predictions = None
        elif ksum_tail:
            from .. import ops
            predictions = ops.pinnsf_epilogue_ksum(ped_msgs, out_obs, self_features, self.tau,
                                                   agent_norm=self_features.dim() == 3 and not self.fix_dest_norm)
        elif FUSED_GLUE and self_features.is_cuda and self_features.dtype == torch.float32 \
                and (self_features.dim() in (2, 3) or self.fix_dest_norm):
            from .. import ops       # one fused kernel; 3-D input without fix_dest_norm keeps the dim=1 quirk (Q2)
            quirk = self_features.dim() == 3 and not self.fix_dest_norm
            if quirk and self._park_tail(acc, acc_o, self_features):
                predictions = None
            else:
                predictions = ops.pinnsf_epilogue(acc, acc_o, self_features, self.tau, agent_norm=quirk)
        else:
            if acc_o is not None:
                acc = acc + acc_o
            predictions = acc + self.desired_force(self_features)
        if self.residual and predictions is not None:
            predictions = predictions + self._correct(encoded)
        out = [predictions, ped_msgs]
        if out_obs is not None:
            out.append(out_obs)
        if self.collision_head is not None and self.predictions_only:
            out.append(None)                    # inference rollouts read out[0] only: the auxiliary head is not launched
        elif self.collision_head is not None:
            src = ped_msgs if self.collision_head == 'msgs' else decoded
            head = self.ped_collision_predictor.mlp
            if self.collision_head == 'decoded' and FUSED_GLUE and FUSED_ROW_DECODER and src.is_cuda \
                    and src.dtype == torch.float32 and src.shape[-1] == 64 and len(head) == 4 \
                    and (head[0].in_features, head[0].out_features, head[2].out_features) == (64, 64, 1) \
                    and isinstance(head[1], nn.ReLU) and isinstance(head[3], nn.Identity) and src.numel() > 0:
                from .. import ops       # `pinnsf_bm`: the head on hand-written kernels, forward and backward
                pc = ops.collision_head64(src, head[0].weight, head[0].bias, head[2].weight, head[2].bias)
                out.append(pc.unsqueeze(-1).squeeze())
            else:
                out.append(torch.sigmoid(self.ped_collision_predictor(src)).squeeze())
        return out


class PINNSF(_PINNSFBase):
    """model.py:720-792 (`--model pinnsf`, and the pre-training net of `pinnsf_res`)."""


class PINNSF_residual(_PINNSFBase):
    """model.py:973-1059 (`--model pinnsf_res`, fine-tune stage)."""
    residual = True
    obs_encoder_in = 0


class PINNSF_bottleneck(_PINNSFBase):
    """model.py:1062-1135 (`--model pinnsf_bottleneck`)."""
    bottleneck = True


class PINNSF_bottleneck_multitask(_PINNSFBase):
    """model.py:1138-1221 (`--model pinnsf_bm`)."""
    bottleneck = True
    collision_head = 'decoded'
    taus = (2, 5 / 6)


class PINNSF_multitask(_PINNSFBase):
    """model.py:1224-1305 (`--model pinnsf_m`, the default)."""
    collision_head = 'msgs'
    taus = (0.5, 5 / 6)


class PINNSF_polar_bottleneck(_PINNSFBase):
    """model.py:1447+ (`--model pinnsf_pb`): the bottleneck network whose per-neighbour outputs are (r, theta)
    in the polar frame of the agent's heading; the two branch sums are rotated back with polar_to_cart
    (src/data/data.py:902-920) before the desired force is added.  tau = 2 for every dataset.
    The heading of `self_features[..., 2:4]` follows Pedestrians.get_heading_direction (data.py:350-395),
    including its temporal fill over dim -3 -- the SLICE axis of channelled (C, N, 7) input.  Deviation: entries
    filled from another frame / slice pass no gradient to that source velocity (they have zero velocity)."""
    bottleneck = True
    taus = (2, 2)
    collision = False

    def __init__(self, args):
        super().__init__(args)
        self.tau = 2
        # the reference reads args.time_unit, which its own main.py never sets (`--model pinnsf_pbc` dies with an
        # AttributeError there); fall back to the 0.08 s of every shipped dataset
        self.time_unit = getattr(args, 'time_unit', 0.08)
        self.collision_threshold = getattr(args, 'collision_threshold', 0.5)

    @staticmethod
    def _heading(velocity):
        n = torch.norm(velocity, p=2, dim=-1, keepdim=True)
        unit = velocity / torch.where(n == 0, n + 0.1, n)
        if velocity.dim() >= 3 and velocity.is_cuda:                       # temporal fill of zero-velocity entries
            from .. import ops
            filled = ops.heading_direction(velocity.detach())
            unit = torch.where(n == 0, filled, unit)
        elif velocity.dim() >= 3:
            raise NotImplementedError('the temporal heading fill of channelled input runs on the GPU only')
        return unit

    @staticmethod
    def _polar_to_cart(points, base):
        """(r, theta) in the frame whose polar axis is `base` (unit vectors) -> (x, y)   (data.py:902-920,
        with cart_to_polar(base, (1, 0)) of :872-900 inlined: sign(0) = 0 makes a heading on the x axis theta = 0)."""
        vol = torch.norm(base, p=2, dim=-1, keepdim=True)
        vol_ = torch.where(vol == 0, vol + 0.1, vol)
        p = base / vol_
        theta_b = torch.acos(torch.clamp(base[..., :1] / vol_, -1 + 1e-6, 1 - 1e-6)) * torch.sign(p[..., 1:2])
        ang = points[..., 1:2] + theta_b
        return torch.cat((points[..., :1] * torch.cos(ang), points[..., :1] * torch.sin(ang)), dim=-1)

    def forward(self, ped_features, obs_features, self_features):
        assert (self_features.shape[-1] == 7), 'Error: PINN model do not accept inputs of historical velocity'
        base = self._heading(self_features[..., -5:-3])

        def branch(feats, encoder, processor, decoder, predictor):
            polar_sum, msgs, _, _ = self._branch(feats, encoder, processor, decoder, predictor)
            if self.collision:       # pinnsf_pbc: sum the polar messages, rotate the sum (model.py:1365-1367)
                return self._polar_to_cart(polar_sum, base), msgs
            cart = self._polar_to_cart(msgs, base.unsqueeze(-2))     # pinnsf_pb: rotate every message (:1507-1510)
            return cart.sum(dim=-2), cart
        acc, ped_msgs = branch(ped_features, self.ped_encoder, self.ped_processor, self.ped_decoder,
                               self.ped_predictor)
        obs_msgs = None
        if self.obs_feature_dim > 0:
            acc_o, obs_msgs = branch(obs_features, self.obs_encoder, self.obs_processor, self.obs_decoder,
                                     self.obs_predictor)
            acc = acc + acc_o
        if FUSED_GLUE and self_features.is_cuda and self_features.dtype == torch.float32 \
                and (self_features.dim() in (2, 3) or self.fix_dest_norm):
            from .. import ops
            quirk = self_features.dim() == 3 and not self.fix_dest_norm
            predictions = ops.pinnsf_epilogue(acc, None, self_features, self.tau, agent_norm=quirk)
        else:
            predictions = acc + self.desired_force(self_features)
        if self.collision:
            from .. import ops
            predictions = ops.collision_post_correction(predictions, ped_features, self_features[..., 2:4],
                                                        self.collision_threshold, self.time_unit)
        out = [predictions, ped_msgs]
        if obs_msgs is not None:
            out.append(obs_msgs)
        return out


class PINNSF_polar_bottleneck_collision(PINNSF_polar_bottleneck):
    """model.py:1307-1444 (`--model pinnsf_pbc`): PINNSF_polar_bottleneck followed by the hand-written collision
    handling on the gathered neighbours (ops.collision_post_correction; GPU only)."""
    collision = True


MODEL_TABLE = {   # simulators.py:40-106 (set_model / set_ft_model), PINNSF-family rows
    'pinnsf': (PINNSF, PINNSF), 'pinnsf_res': (PINNSF, PINNSF_residual),
    'pinnsf_bottleneck': (PINNSF_bottleneck, PINNSF_bottleneck),
    'pinnsf_bm': (PINNSF_bottleneck_multitask, PINNSF_bottleneck_multitask),
    'pinnsf_m': (PINNSF_multitask, PINNSF_multitask),
    'pinnsf_pb': (PINNSF_polar_bottleneck, PINNSF_polar_bottleneck),
    'pinnsf_pbc': (PINNSF_polar_bottleneck_collision, PINNSF_polar_bottleneck_collision),
}

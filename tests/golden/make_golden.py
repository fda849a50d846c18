#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING the real reference.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box); the .npz files it writes next to this script are committed and are what the tests
read.  No reference source is copied: the reference is imported, called and its outputs
are saved.  Usage:  python tests/golden/make_golden.py [names...]

Fixture families:
  relfeat_*    Pedestrians.get_relative_features (+ get_nearby_obj_in_sight indices)
  collision_*  Pedestrians.collision_detection / calculate_collision_label
  mlapm_*      MLAPM.step (raw / GC as shipped; UCY with the one-line coll.unsqueeze(-1)
               fix applied at import time) + autograd gradients
  calcacc      utils.calc_acceleration
  model_*      PINNSF family forward outputs + state_dicts (seed 666)
  model_polar  `--model pinnsf_pb` / `pinnsf_pbc`: outputs, the acceleration before the hand-written collision
               correction (pins SURVEY row a9 in isolation), gradients on a dense scene
  dataset      TimeIndexedPedData.make_dataset / channelled windows on the shipped clips
  rollout*     BaseSimulator.get_multiple_rollouts / test_multiple_rollouts_for_training: positions, the seven
               loss scalars, collision counts, gradients (rollout_more: UCY + pinnsf_res; rollout_flags: the
               non-default loss switches)
  metrics      functions.metrics (MAE, OT, MMD, collisions)
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, os.path.join(REF, 'src'))
sys.path.insert(0, REPO)
sys.modules.setdefault('setproctitle', types.SimpleNamespace(setproctitle=lambda *_: None))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import data.data as DATA  # noqa: E402  (reference)
from piml_amd.scenes import synthetic_gc_scene  # noqa: E402

torch.set_num_threads(8)
GC_CLIP = os.path.join(REF, 'data/GC_Dataset/GC_Dataset_ped1-12685_time1000-1060_interp9_xrange5-25_yrange15-35.npy')
UCY_CLIP = os.path.join(REF, 'data/UCY_dataset/UCY_Dataset_time0-54_timeunit0.08.npy')
TOY = os.path.join(REF, 'data/GC_Dataset/GC_Dataset_toy1.npy')


def T(x):
    return torch.tensor(np.asarray(x))


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f'wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)')


def load_raw(path):
    raw = DATA.RawData()
    raw.load_trajectory_data(path)
    return raw


def run_relfeat(p, v, a, dest, obs, kp=6, ang_p=90, dp=4, ko=10, ang_o=90, do=4):
    """Call the reference; also capture the neighbour indices/distances it used."""
    ped = DATA.Pedestrians()
    p, v, a, dest, obs = [T(x).clone() for x in (p, v, a, dest, obs)]
    v_in, a_in = v.clone(), a.clone()
    pf, of, df = ped.get_relative_features(p, v, a, dest, obs, kp, ang_p, dp, ko, ang_o, do)
    hd = ped.get_heading_direction(v)
    pd_, pi_ = ped.get_nearby_obj_in_sight(p, p, hd, kp, ang_p)
    dim = obs.dim()
    obs_t = obs.unsqueeze(-3).repeat(*([1] * (dim - 2) + [p.shape[-3]] + [1, 1]))
    if p.dim() > obs_t.dim():
        obs_t = obs_t.expand(*p.shape[:-2], *obs_t.shape[-2:])
    od_, oi_ = ped.get_nearby_obj_in_sight(p, obs_t, hd, ko, ang_o)
    return dict(position=p, velocity=v_in, acceleration=a_in, destination=dest, obstacles=obs,
                params=np.array([kp, ang_p, dp, ko, ang_o, do], np.float64),
                ped_features=pf, obs_features=of, dest_features=df, heading=hd,
                velocity_after=v, acceleration_after=a,
                ped_idx=pi_.to(torch.int32), ped_dist=pd_, obs_idx=oi_.to(torch.int32), obs_dist=od_)


def gen_relfeat():
    # (i) real clips: single frames (t=1, the per-step call) and a short multi-frame window
    for tag, path in (('gc', GC_CLIP), ('ucy', UCY_CLIP)):
        raw = load_raw(path)
        frames = [25, 100, 400, min(749, raw.num_steps - 1)]
        for f in frames:
            sl = slice(f, f + 1)
            save(f'relfeat_{tag}_f{f}', **run_relfeat(raw.position[sl], raw.velocity[sl], raw.acceleration[sl],
                                                     raw.destination[sl], raw.obstacles))
        sl = slice(96, 104)
        save(f'relfeat_{tag}_window', **run_relfeat(raw.position[sl], raw.velocity[sl], raw.acceleration[sl],
                                                   raw.destination[sl], raw.obstacles))
    raw = load_raw(TOY)
    save('relfeat_toy1', **run_relfeat(raw.position, raw.velocity, raw.acceleration, raw.destination, raw.obstacles))

    # (iii) seeded synthetic scenes
    for N, M, seed, ang in ((64, 0, 0, 90), (64, 100, 1, 100), (256, 100, 2, 90), (256, 2000, 0, 90),
                            (1024, 100, 1, 90), (1024, 2000, 2, 90)):
        sc = synthetic_gc_scene(N, M, seed=seed)
        rng = np.random.default_rng(1000 + seed)
        v = sc['velocity'].copy()
        v[rng.random(N) < 0.05] = 0.0            # standing agents: heading 0 -> blind
        a = (rng.standard_normal((N, 2)) * 0.3).astype(np.float32)
        a[rng.random(N) < 0.02] = np.nan          # exercised by the NaN->0 in-place rule
        save(f'relfeat_syn_N{N}_M{M}_a{ang}',
             **run_relfeat(sc['position'][None], v[None], a[None], sc['destination'][None], sc['obstacles'],
                           ang_p=ang, ang_o=ang))
    # channelled (C,1,N,2) input as HOT LOOP C uses it (simulators.py:772-776)
    sc = synthetic_gc_scene(96, 100, seed=5, channels=4)
    save('relfeat_syn_channels',
         **run_relfeat(sc['position'][:, None], sc['velocity'][:, None], sc['acceleration'][:, None],
                       sc['destination'][:, None], sc['obstacles']))
    # non-default k / thresholds, k > #objects
    sc = synthetic_gc_scene(40, 0, seed=7)
    save('relfeat_syn_smallk', **run_relfeat(sc['position'][None], sc['velocity'][None], sc['acceleration'][None],
                                             sc['destination'][None], sc['obstacles'], kp=3, dp=2, ko=10, do=3))
    sc = synthetic_gc_scene(5, 100, seed=8, nan_frac=0.0)
    save('relfeat_syn_kgtN', **run_relfeat(sc['position'][None], sc['velocity'][None], sc['acceleration'][None],
                                           sc['destination'][None], sc['obstacles'], kp=6, ko=10))


def gen_collision():
    raw = load_raw(GC_CLIP)
    ped = DATA.Pedestrians
    p3 = raw.position[380:440].clone()               # (t,N,2) with NaNs, > 25 frames for the friends rule
    out = {}
    for thr in (0.5, 0.25, 1.5):
        out[f'coll3_thr{thr}'] = ped.collision_detection(p3.clone(), thr).to(torch.uint8)
    out['coll3_real_thr0.5'] = ped.collision_detection(p3.clone() + 0.05, 0.5, real_position=p3.clone()).to(torch.uint8)
    out['coll3_real_thr1.5'] = ped.collision_detection(p3.clone() + 0.05, 1.5, real_position=p3.clone()).to(torch.uint8)
    p4 = torch.stack([raw.position[s:s + 6] for s in (100, 250, 400, 560)])   # (C,T,N,2)
    for thr in (0.5, 1.5):
        out[f'coll4_thr{thr}'] = ped.collision_detection(p4.clone(), thr).to(torch.uint8)
    pc = raw.position[[100, 250, 400, 560]].clone()  # (C,N,2) as in HOT LOOP C (simulators.py:708)
    for thr in (0.5, 0.25, 1.5):
        out[f'collc_thr{thr}'] = ped.collision_detection(pc.clone(), thr).to(torch.uint8)
    save('collision_gc', p3=p3, p4=p4, pc=pc, **out)

    sc = synthetic_gc_scene(512, 100, seed=3, channels=3)
    ps = T(sc['position'])
    save('collision_syn', pc=ps, **{f'collc_thr{thr}': ped.collision_detection(ps.clone(), thr).to(torch.uint8)
                                    for thr in (0.5, 0.25)})

    # collision label on real gathered features
    g = np.load(os.path.join(HERE, 'relfeat_gc_f400.npz'))
    pf = T(g['ped_features'])
    rng = np.random.default_rng(11)
    rnd = T((rng.standard_normal((64, 6, 6)) * np.array([0.6, 0.6, 1.2, 1.2, 1, 1])).astype(np.float32))
    rnd[0, 0] = 0.0
    save('collision_label', feat_real=pf, label_real=ped.calculate_collision_label(pf.clone()),
         feat_rnd=rnd, label_rnd=ped.calculate_collision_label(rnd.clone()))


def _mlapm_class(fixed_ucy):
    """The reference class; for UCY the documented one-line fix is applied to the module
    text at import time (nothing is written to disk)."""
    path = os.path.join(REF, 'src/models/mlapm.py')
    src = open(path).read()
    if fixed_ucy:
        old = "self.args['B'] * r * coll + self.args['C'] * coll"
        assert old in src
        src = src.replace(old, "self.args['B'] * r * coll.unsqueeze(-1) + self.args['C'] * coll.unsqueeze(-1)")
    mod = types.ModuleType('ref_mlapm')
    exec(compile(src, path, 'exec'), mod.__dict__)
    return mod.MLAPM


def gen_mlapm():
    params = {
        'raw': dict(version='raw', tau=0.5, A=7.55, B=-3.00),
        'GC': dict(version='GC', tau=0.5, A=7.55, B=-3.00, C=0.2, D=-0.3, theta=56),   # main_mlapm.py:16
        'UCY': dict(version='UCY', tau=5 / 6, A=10.67, B=-3.33, C=0.5, theta=20),
    }
    out = {}
    for ver, pr in params.items():
        M = _mlapm_class(ver == 'UCY')(**pr)
        for N in (7, 64, 1024):
            if N == 7:   # the main_mlapm.py:8-14 antipodal circle
                th = torch.linspace(0, 2 * torch.pi * (1 - 1. / N), N)
                p = torch.stack([10 * th.cos(), 10 * th.sin()], dim=-1)
                g = torch.Generator().manual_seed(0)
                v = torch.rand(N, 2, generator=g)
                v0 = torch.full([N, 1], 1.5)
                d = -p
            else:
                sc = synthetic_gc_scene(N, 0, seed=N, nan_frac=0.0)
                p, v, d, v0 = (T(sc[k]) for k in ('position', 'velocity', 'destination', 'desired_speed'))
                if N == 64:
                    p = p * 0.35 + 5.0      # dense: collisions / tmin branches fire
                rng = np.random.default_rng(N)
                v = v + T((rng.standard_normal((N, 2)) * 0.4).astype(np.float32))
            pg, vg, v0g, dg = [x.clone().requires_grad_(True) for x in (p, v, v0, d)]
            act = M.step(pg, vg, v0g, dg, dt=0.08, radius=0.3)
            gen = torch.Generator().manual_seed(N)
            w = torch.randn(act.shape, generator=gen)
            gp, gv, gv0, gd = torch.autograd.grad((act * w).sum(), (pg, vg, v0g, dg))
            key = f'{ver}_N{N}'
            out.update({f'{key}_p': p, f'{key}_v': v, f'{key}_v0': v0, f'{key}_dest': d,
                        f'{key}_action': act, f'{key}_w': w, f'{key}_gp': gp, f'{key}_gv': gv,
                        f'{key}_gv0': gv0, f'{key}_gdest': gd})
        out[f'{ver}_params'] = np.array([pr.get(k, 0.0) for k in ('tau', 'A', 'B', 'C', 'D', 'theta')], np.float64)
    # the 200-step demo trajectory of main_mlapm.py:18-36 (fixed velocity seed)
    M = _mlapm_class(False)(**params['GC'])
    N = 7
    th = torch.linspace(0, 2 * torch.pi * (1 - 1. / N), N)
    p = torch.stack([10 * th.cos(), 10 * th.sin()], dim=-1)
    v = torch.rand(N, 2, generator=torch.Generator().manual_seed(0))
    v0 = torch.full([N, 1], 1.5)
    d = -p.clone()
    traj = [p.clone()]
    for _ in range(200):
        v = M.step(p, v, v0, d, dt=0.08, radius=0.3)
        p = p + v * 0.08
        traj.append(p.clone())
    out['demo_traj'] = torch.stack(traj)
    save('mlapm', **out)


def gen_calcacc():
    import utils.utils as UTILS
    g = np.load(os.path.join(HERE, 'relfeat_gc_f400.npz'))
    pf = T(g['ped_features'])                                  # (1,N,k,6)
    rng = np.random.default_rng(5)
    rnd = T((rng.standard_normal((3, 50, 6, 6))).astype(np.float32))
    out = {'feat_real': pf, 'feat_rnd': rnd}
    for ver, ds in (('v0', 'gc1560'), ('v0', 'ucy'), ('v1', 'ucy'), ('v2', 'gc2344')):
        out[f'real_{ver}_{ds}'] = UTILS.calc_acceleration(pf[0].clone(), ver, ds)
        out[f'rnd_{ver}_{ds}'] = UTILS.calc_acceleration(rnd.clone(), ver, ds)
    save('calcacc', **out)


def model_args(**kw):
    a = types.SimpleNamespace(
        ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=128,
        processor_hidden_size=128, decoder_hidden_size=64, encoder_hidden_layers=3,
        processor_hidden_layers=16, decoder_hidden_layers=2, dropout=0.5, activation='relu',
        dataset_name='ucy', res_hidden_layers=3, correction_hidden_layers=1, time_unit=0.08,
        collision_threshold=0.5)
    a.__dict__.update(kw)
    return a


def gen_model():
    import models.model as MODEL
    g = np.load(os.path.join(HERE, 'relfeat_gc_f400.npz'))
    pf, of, df = T(g['ped_features'])[0], T(g['obs_features'])[0], T(g['dest_features'])[0]   # (N,k,6)
    N = pf.shape[0]
    rng = np.random.default_rng(9)
    selff = torch.cat((df, T(g['velocity'])[0], T(np.nan_to_num(g['acceleration']))[0],
                       T((1.0 + rng.random((N, 1))).astype(np.float32))), dim=-1)            # (N,7)
    gc = np.load(os.path.join(HERE, 'relfeat_syn_channels.npz'))
    pfc, ofc, dfc = T(gc['ped_features'])[:, 0], T(gc['obs_features'])[:, 0], T(gc['dest_features'])[:, 0]
    selfc = torch.cat((dfc, T(gc['velocity'])[:, 0], T(gc['acceleration'])[:, 0],
                       T((1.0 + rng.random(dfc.shape[:-1] + (1,))).astype(np.float32))), dim=-1)
    out = dict(ped=pf, obs=of, selff=selff, pedc=pfc, obsc=ofc, selfc=selfc)
    for name, cls, kw in (('pinnsf_m', MODEL.PINNSF_multitask, {}),
                          ('pinnsf_m_gc', MODEL.PINNSF_multitask, dict(dataset_name='gc1560')),
                          ('pinnsf_bm', MODEL.PINNSF_bottleneck_multitask, {}),
                          ('pinnsf', MODEL.PINNSF, {}),
                          ('pinnsf_bottleneck', MODEL.PINNSF_bottleneck, {}),
                          ('pinnsf_res', MODEL.PINNSF_residual, {}),
                          ('pinnsf_m_p1', MODEL.PINNSF_multitask, dict(processor_hidden_layers=1))):
        torch.manual_seed(666)
        m = cls(model_args(**kw)).eval()
        for k, v in m.state_dict().items():
            out[f'{name}/sd/{k}'] = v
        with torch.no_grad():
            for tag, args in (('n', (pf, of, selff)), ('c', (pfc, ofc, selfc))):
                res = m(*[x.clone() for x in args])
                for q, r in enumerate(res):
                    out[f'{name}/out_{tag}{q}'] = r
    save('model', **out)


def gen_model_polar():
    """`--model pinnsf_pb` / `pinnsf_pbc` (src/models/model.py:1307-1460+): polar bottleneck network and the
    hand-written collision post-correction.  Inputs: the real GC frame and the channelled
    synthetic case of `model.npz`, plus a DENSE synthetic scene (many neighbours inside the reaction radius)
    whose features come from the reference's own get_relative_features; for that case also the gradients of a
    weighted sum of the corrected acceleration w.r.t. the three inputs and two weight tensors."""
    import models.model as MODEL
    gm = np.load(os.path.join(HERE, 'model.npz'))
    cases = {'n': tuple(T(gm[k]) for k in ('ped', 'obs', 'selff')),
             'c': tuple(T(gm[k]) for k in ('pedc', 'obsc', 'selfc'))}
    sc = synthetic_gc_scene(300, 100, seed=11)
    rng = np.random.default_rng(12)
    pos = (sc['position'] * 0.35).astype(np.float32)                       # ~8x the density: plenty of near misses
    vel = (sc['velocity'] + rng.standard_normal(sc['velocity'].shape) * 0.4).astype(np.float32)
    acc = (rng.standard_normal(pos.shape) * 0.3).astype(np.float32)
    dest = (sc['destination'] * 0.35).astype(np.float32)
    obs = (sc['obstacles'] * 0.35).astype(np.float32)
    P = DATA.Pedestrians()
    pf, of, df = P.get_relative_features(T(pos)[None], T(np.nan_to_num(vel))[None], T(acc)[None], T(dest)[None],
                                         T(obs), 6, 90, 4, 10, 90, 4)
    selfd = torch.cat((df[0], T(np.nan_to_num(vel)), T(acc), T(sc['desired_speed'])), dim=-1)
    cases['d'] = (pf[0], of[0], selfd)
    out = {}
    for tag, (a, b, c) in cases.items():
        out[f'in_{tag}/ped'], out[f'in_{tag}/obs'], out[f'in_{tag}/selff'] = a, b, c
    for name, cls in (('pinnsf_pb', MODEL.PINNSF_polar_bottleneck), ('pinnsf_pbc', MODEL.PINNSF_polar_bottleneck_collision)):
        torch.manual_seed(666)
        m = cls(model_args()).eval()
        for k, v in m.state_dict().items():
            out[f'{name}/sd/{k}'] = v
        with torch.no_grad():
            for tag, args in cases.items():
                res = m(*[x.clone() for x in args])
                for q, r in enumerate(res):
                    out[f'{name}/out_{tag}{q}'] = r
                if name == 'pinnsf_pbc':
                    # the acceleration BEFORE the hand-written correction, from the reference's own submodules
                    # and coordinate helpers (model.py:1355-1381): pins the correction in isolation
                    ped, obs, sf = [x.clone() for x in args]
                    base = DATA.Pedestrians.get_heading_direction(sf[..., -5:-3])
                    e = m.ped_predictor(m.ped_decoder(m.ped_processor(m.ped_encoder(ped)))).sum(-2)
                    acc = DATA.TimeIndexedPedDataPolarCoor.polar_to_cart(e, base)
                    e2 = m.obs_predictor(m.obs_decoder(m.obs_processor(m.obs_encoder(obs)))).sum(-2)
                    acc = acc + DATA.TimeIndexedPedDataPolarCoor.polar_to_cart(e2, base)
                    t_ = torch.norm(sf[..., :2], p=2, dim=1, keepdim=True)
                    t_ = torch.where(t_ == 0, t_ + 0.1, t_)
                    out[f'{name}/pre_{tag}'] = acc + (sf[..., -1:] * (sf[..., :2] / t_) - sf[..., 2:4]) / m.tau
        ins = [x.clone().requires_grad_(True) for x in cases['d']]
        m.zero_grad()
        res = m(*ins)
        w = torch.linspace(-1.0, 1.0, res[0].numel()).view_as(res[0])
        (res[0] * w).sum().backward()
        for k, x in zip(('ped', 'obs', 'selff'), ins):
            out[f'{name}/grad_d/{k}'] = torch.nan_to_num(x.grad)
        for k, p in m.named_parameters():
            if k in ('ped_predictor.mlp.0.weight', 'obs_encoder.mlp.0.weight', 'ped_encoder.mlp.4.bias'):
                out[f'{name}/grad_d/param/{k}'] = p.grad.clone()
    save('model_polar', **out)


def sim_args(**kw):
    """The argparse namespace of src/main.py:26-112 (defaults), as far as the simulator reads it."""
    a = model_args(dataset_name='gc1560')
    a.__dict__.update(
        model='pinnsf_m', device='cpu', gpus='3', learning_rate=0.002, weight_decay=5e-4, batch_size=3,
        topk_ped=6, topk_obs=10, sight_angle_ped=90, sight_angle_obs=90, dist_threshold_ped=4,
        dist_threshold_obs=4, num_history_velocity=1, skip_frames=25, valid_steps=5, time_decay=1,
        reg_weight=0., collision_threshold=0.5, collision_loss_weight=10, val_coll_weight=30,
        hard_collision_penalty=10, teacher_weight=0, collision_pred_weight=10, collision_focus_weight=10,
        new_collision_loss_flag=0, collision_loss_version='v0', finetune_lr_decay=1, finetune_wd_aug=1,
        ft_lr_decay2=0., exp_name='golden', model_name_suffix='x', epochs=1, patience=1, ft_patience=5)
    a.__dict__.update(kw)
    return a


DATA_FIELDS = ('ped_features', 'obs_features', 'self_features', 'labels', 'position', 'velocity', 'acceleration',
               'destination', 'dest_idx', 'dest_num', 'waypoints', 'obstacles', 'mask_p', 'mask_p_pred',
               'abnormal_mask')


def dump_data(prefix, d, out):
    for k in DATA_FIELDS:
        out[f'{prefix}/{k}'] = getattr(d, k).clone()
    out[f'{prefix}/time_unit'] = np.float64(d.time_unit)
    out[f'{prefix}/num_frames'] = np.int64(d.num_frames)


def gen_rollout():
    import models.simulators as SIM
    out = {}
    raw = load_raw(GC_CLIP)
    # ---- HOT LOOP B: inference rollout (simulators.py:556-657) on frames 300..459 of the clip ----
    sl = list(range(300, 460))
    args = sim_args()
    full = DATA.TimeIndexedPedData()
    full.make_dataset(args, raw)
    full.set_dataset_info(full, raw, list(range(len(full))))
    data = DATA.TimeIndexedPedData()
    for k in ('ped_features', 'obs_features', 'self_features', 'labels'):
        setattr(data, k, getattr(full, k)[sl].clone())
    data.topk_obs = full.topk_obs
    data.num_frames = len(sl)
    data.set_dataset_info(full, raw, sl)
    data.num_frames = data.dataset_len = len(sl)
    for k in ('position', 'velocity', 'acceleration', 'destination', 'dest_idx', 'mask_p', 'mask_p_pred'):
        setattr(data, k, getattr(data, k).clone())
    data.mask_p_pred = data.mask_p.clone()          # every present agent is simulated
    data.mask_p_pred[:1] = data.mask_p[:1]
    torch.manual_seed(666)
    sim = SIM.BaseSimulator(args)
    sim.model.eval()
    for k, v in sim.model.state_dict().items():
        out[f'sd_m/{k}'] = v.clone()
    dump_data('roll', data, out)
    with torch.no_grad():
        res = sim.get_multiple_rollouts(data, t_start=0, load_model=False)
    out['roll/out_position'] = res.position
    out['roll/out_velocity'] = res.velocity
    out['roll/out_acceleration'] = res.acceleration
    out['roll/out_mask_p'] = res.mask_p

    # ---- HOT LOOP C: differentiable training rollout (simulators.py:659-832), C = 4 windows ----
    for model_name in ('pinnsf_m', 'pinnsf_bm'):
        args = sim_args(model=model_name)
        cdata = full.to_channeled_time_index_data(args.valid_steps, 'slice')
        batch = DATA.ChanneledTimeIndexedPedData.slice(cdata, [396, 401, 406, 411])
        for k in DATA_FIELDS:
            if torch.is_tensor(getattr(batch, k)):
                setattr(batch, k, getattr(batch, k).clone())
        torch.manual_seed(666)
        sim = SIM.BaseSimulator(args)
        sim.model.eval()
        sim.collision_count = sim.hard_collision_count = 0
        sim.epoch = sim.batch_idx = 0
        tag = f'train_{model_name}'
        for k, v in sim.model.state_dict().items():
            out[f'{tag}/sd/{k}'] = v.clone()
        dump_data(tag, batch, out)
        res = sim.test_multiple_rollouts_for_training(batch)
        res[0].backward()
        out[f'{tag}/scalars'] = np.array([float(x) for x in res], np.float64)
        out[f'{tag}/counts'] = np.array([sim.collision_count, sim.hard_collision_count], np.float64)
        for k, p in sim.model.named_parameters():
            out[f'{tag}/grad/{k}'] = torch.zeros_like(p) if p.grad is None else p.grad.clone()
    save('rollout', **out)


def gen_dataset():
    """The feature-building pipeline (data.py:746-833, 958-1043, 1046-1123) on the toy clip (whole)
    and on the real GC clip (every 25th frame kept, to bound the fixture size)."""
    out = {}
    args = sim_args()
    for tag, path, keep in (('toy1', TOY, 1), ('gc', GC_CLIP, 25)):
        raw = load_raw(path)
        d = DATA.TimeIndexedPedData()
        d.make_dataset(args, raw)
        d.set_dataset_info(d, raw, list(range(len(d))))
        for k in ('ped_features', 'obs_features', 'self_features', 'labels', 'mask_p_pred', 'mask_a_pred', 'mask_v_pred'):
            out[f'{tag}/{k}'] = getattr(d, k)[::keep].clone()
        out[f'{tag}/abnormal_mask'] = d.abnormal_mask.clone()
        out[f'{tag}/desired_speed'] = d.self_features[0, :, -1].clone()
        for k in ('position', 'velocity', 'acceleration', 'destination', 'dest_idx', 'mask_p', 'mask_v', 'mask_a'):
            out[f'{tag}/raw_{k}'] = getattr(raw, k)[::keep].clone()
        out[f'{tag}/raw_waypoints'] = raw.waypoints.clone()
        out[f'{tag}/raw_dest_num'] = raw.dest_num.clone()
        out[f'{tag}/raw_obstacles'] = raw.obstacles.clone()
        ch = d.to_channeled_time_index_data(args.valid_steps, 'slice')
        out[f'{tag}/ch_slice_shape'] = np.array(ch.position.shape)
        out[f'{tag}/ch_slice_pos_win7'] = ch.position[7].clone()
        out[f'{tag}/ch_slice_pf_sum'] = ch.ped_features.sum(dim=(1, 2, 3, 4)).clone()
        sp = d.to_channeled_time_index_data(args.valid_steps, 'split')
        out[f'{tag}/ch_split_shape'] = np.array(sp.position.shape)
        out[f'{tag}/ch_split_pos_win3'] = sp.position[3].clone()
        pw = d.to_pointwise_data()        # NB: shifts d.labels in place in the reference -> call last
        out[f'{tag}/pw_len'] = np.int64(len(pw))
        out[f'{tag}/pw_labels_head'] = pw.labels[:64].clone()
        out[f'{tag}/pw_self_head'] = pw.self_features[:64].clone()
        out[f'{tag}/pw_ped_sum'] = pw.ped_features.sum(dim=(1, 2)).clone()
    save('dataset', **out)


def gen_metrics():
    """Evaluation metrics (metrics.py:16-273) on a perturbed copy of real GC frames."""
    import functions.metrics as METRIC
    raw = load_raw(GC_CLIP)
    q = raw.position[380:440].clone()
    g = torch.Generator().manual_seed(3)
    p = q + 0.3 * torch.randn(q.shape, generator=g)
    mask = raw.mask_p[380:440].clone()
    out = dict(p=p, q=q, mask=mask)
    for red in ('sum', 'mean'):
        out[f'ot_{red}'] = np.float64(METRIC.ot_with_time_mask(p, q, mask, reduction=red))
        out[f'mmd_{red}'] = np.float64(METRIC.mmd_with_time_mask(p, q, mask, reduction=red))
        out[f'mae_{red}'] = np.float64(METRIC.mae_with_time_mask(p, q, mask, reduction=red))
        out[f'coll_{red}'] = np.float64(METRIC.collision_count(q, 0.6, reduction=red))
    save('metrics', **out)


def _train_case(out, tag, path, model_name, ds, wins, valid_steps=6, **overrides):
    """One batch of rollout windows through the reference's test_multiple_rollouts_for_training."""
    import models.simulators as SIM
    raw = load_raw(path)
    args = sim_args(model=model_name, dataset_name=ds, valid_steps=valid_steps, **overrides)
    full = DATA.TimeIndexedPedData()
    full.make_dataset(args, raw)
    full.set_dataset_info(full, raw, list(range(len(full))))
    cdata = full.to_channeled_time_index_data(args.valid_steps, 'slice')
    batch = DATA.ChanneledTimeIndexedPedData.slice(cdata, wins)
    for k in DATA_FIELDS:
        if torch.is_tensor(getattr(batch, k)):
            setattr(batch, k, getattr(batch, k).clone())
    torch.manual_seed(666)
    sim = SIM.BaseSimulator(args)
    if model_name == 'pinnsf_res':
        torch.manual_seed(667)
        sim.set_ft_model(args)
    sim.model.eval()
    sim.collision_count = sim.hard_collision_count = 0
    sim.epoch = sim.batch_idx = 0
    for k, v in sim.model.state_dict().items():
        out[f'{tag}/sd/{k}'] = v.clone()
    dump_data(tag, batch, out)
    res = sim.test_multiple_rollouts_for_training(batch)
    res[0].backward()
    out[f'{tag}/scalars'] = np.array([float(x.detach()) for x in res], np.float64)
    out[f'{tag}/counts'] = np.array([sim.collision_count, sim.hard_collision_count], np.float64)
    gsum = {k: (torch.zeros(()) if p.grad is None else p.grad.abs().sum()) for k, p in sim.model.named_parameters()}
    for k, p in sim.model.named_parameters():
        if k.startswith(('ped_encoder.mlp.0', 'obs_encoder.mlp.4', 'ped_predictor', 'corrector.2')):
            out[f'{tag}/grad/{k}'] = torch.zeros_like(p) if p.grad is None else p.grad.clone()
    out[f'{tag}/grad_abs_sum'] = np.float64(sum(float(v) for v in gsum.values()))


def gen_rollout_more():
    """More HOT LOOP C cases: the UCY configuration (tau = 5/6, no obstacles -> 2-point placeholder) and
    the residual fine-tune network (`--model pinnsf_res` after set_ft_model)."""
    out = {}
    _train_case(out, 'ucy_m', UCY_CLIP, 'pinnsf_m', 'ucy', [100, 230, 360, 500])
    _train_case(out, 'gc_res', GC_CLIP, 'pinnsf_res', 'gc1560', [300, 420])
    save('rollout_more', **out)


def gen_rollout_flags():
    """HOT LOOP C with the non-default loss switches on: new_collision_loss_flag (label collisions mask
    the collision terms), teacher_weight (acceleration MSE), reg_weight (cumulative L1 of the messages), and the
    bottleneck-multitask collision head (BCE against calculate_collision_label)."""
    out = {}
    _train_case(out, 'gc_flags_bm', GC_CLIP, 'pinnsf_bm', 'gc1560', [300, 420, 560], new_collision_loss_flag=1,
                teacher_weight=0.5, reg_weight=1e-3)
    _train_case(out, 'gc_flags_m', GC_CLIP, 'pinnsf_m', 'gc1560', [250, 480], new_collision_loss_flag=1,
                teacher_weight=0.25, reg_weight=1e-4, collision_loss_version='v0')
    # the loss switches of the shipped UCY experiment (src/configs/exp_configs/piml-ucydata.yaml): collision loss v2
    # (abnormal-agent mask), time decay 0.9, 10-frame windows, message regulariser, bottleneck collision head
    _train_case(out, 'ucy_exp_bm', UCY_CLIP, 'pinnsf_bm', 'ucy', [120, 400], valid_steps=10,
                collision_loss_version='v2', time_decay=0.9, reg_weight=1e-2, collision_pred_weight=5e-2,
                collision_focus_weight=1, collision_loss_weight=40, hard_collision_penalty=1)
    # ... and of the shipped GC experiment (piml-gcdata.yaml: dataset gc2344, heavier collision weights)
    _train_case(out, 'gc_exp_bm', GC_CLIP, 'pinnsf_bm', 'gc2344', [200, 450], valid_steps=10,
                collision_loss_version='v2', time_decay=0.9, reg_weight=1e-2, collision_pred_weight=5e-2,
                collision_focus_weight=1, collision_loss_weight=200, hard_collision_penalty=2)
    save('rollout_flags', **out)


# ---- cfg5 (BASELINE.json configs[4]): the full pre-train -> fine-tune -> test loop of src/main.py:126-173 ----
MAINFLOW_CASES = {
    # flag sets of the shipped experiments (src/configs/exp_configs/piml-gcdata.yaml / piml-ucydata.yaml), with
    # dropout 0 (deterministic; the RNG stream of dropout cannot be reproduced across devices) and 2 + 2 epochs
    'gc': dict(
        dataset_name='gc2344', collision_loss_weight=200, hard_collision_penalty=2, val_coll_weight=30,
        pretrain=dict(train=['data/synthetic_data/GC_Dataset_ped1-12685_time1000-1060_interp9_xrange5-25_yrange15-35_simulation.npy'],
                      valid=['data/synthetic_data/GC_Dataset_ped1-12685_time1100-1160_interp9_xrange5-25_yrange15-35_simulation.npy']),
        finetune=dict(train=['data/GC_Dataset/GC_Dataset_ped1-12685_time1000-1060_interp9_xrange5-25_yrange15-35.npy'],
                      valid=['data/synthetic_data/GC_Dataset_ped1-12685_time1100-1160_interp9_xrange5-25_yrange15-35_simulation.npy'],
                      test=['data/synthetic_data/GC_Dataset_ped1-12685_time1000-1060_interp9_xrange5-25_yrange15-35_simulation.npy'])),
    'ucy': dict(
        dataset_name='ucy', collision_loss_weight=40, hard_collision_penalty=1, val_coll_weight=10,
        pretrain=dict(train=['data/synthetic_data/UCY_Dataset_time108-162_timeunit0.08_simulation.npy'],
                      valid=['data/synthetic_data/UCY_Dataset_time162-216_timeunit0.08_simulation.npy']),
        # valid / test on the REAL clip (no obstacles -> the 2-point placeholder, k_o = 2).  The synthetic UCY clips carry a
        # single obstacle point: k_o = 1, and the reference's `.squeeze()` calls in get_multiple_rollouts
        # (simulators.py:647-649) then drop the neighbour axis, its bottleneck model sums the obstacle messages over the
        # AGENTS (dim=-2 of an (N, 2) tensor) and every rollout blows up (MSE ~300 m^2) -- SURVEY quirk Q13, not reproduced.
        finetune=dict(train=['data/UCY_dataset/UCY_Dataset_time162-216_timeunit0.08.npy'],
                      valid=['data/UCY_dataset/UCY_Dataset_time162-216_timeunit0.08.npy'],
                      test=['data/UCY_dataset/UCY_Dataset_time162-216_timeunit0.08.npy'])),
}


def mainflow_args(case, **kw):
    c = MAINFLOW_CASES[case]
    a = sim_args(model='pinnsf_bm', dataset_name=c['dataset_name'], dropout=0.0, learning_rate=2e-4, finetune_lr_decay=0.02,
                 batch_size=128, ft_batch_size=32, weight_decay=1e-6, valid_steps=10, time_decay=0.9, epochs=2,
                 collision_loss_version='v2', collision_pred_weight=5e-2, reg_weight=1e-2, teacher_weight=0,
                 true_label_weight=0, collision_focus_weight=1, collision_loss_weight=c['collision_loss_weight'],
                 hard_collision_penalty=c['hard_collision_penalty'], val_coll_weight=c['val_coll_weight'],
                 patience=25, ft_patience=5, finetune_flag=True, pinnsf_interaction='sim', iter_flag=0, seed=666,
                 shuffle=False, exp_name='mainflow_' + case, model_name_suffix='golden', training_mode='normal')
    a.__dict__.update(kw)
    return a


def gen_mainflow(cases=('gc', 'ucy')):
    """Runs the reference's train -> finetune -> test_multiple_rollouts sequence (the calls of src/main.py:126-173,
    with `simulator.finetune` on TimeIndexedPedDataset2 windows because main.py:153 reads an undefined
    args.f_batch_size) and records every number it prints plus the final test rollout."""
    import contextlib
    import io
    import re
    import shutil
    import yaml
    import data.dataset as DATASET
    import models.simulators as SIM
    import utils.data_loader as LOADER
    import functions.metrics as METRIC
    from piml_amd.functions.metrics import fde_with_time_mask
    scratch = '/tmp/piml_ref_mainflow'
    os.makedirs(os.path.join(scratch, 'src'), exist_ok=True)
    cwd = os.getcwd()
    os.chdir(os.path.join(scratch, 'src'))          # the reference writes checkpoints to ../saved_model
    float_re = r'([-+0-9.eE]+|nan|inf)'
    try:
        for case in cases:
            out = {}
            cfg = MAINFLOW_CASES[case]
            args = mainflow_args(case)
            yamls = {}
            for stage in ('pretrain', 'finetune'):
                path = os.path.join(scratch, f'{case}_{stage}.yaml')
                yaml.safe_dump({k: [os.path.join(REF, f) for f in v] for k, v in cfg[stage].items()}, open(path, 'w'))
                yamls[stage] = path
            log = io.StringIO()

            class Tee(io.TextIOBase):
                def write(self, x):
                    log.write(x)
                    sys.__stdout__.write(x)
                    return len(x)
            with contextlib.redirect_stdout(Tee()):
                np.random.seed(args.seed)
                torch.manual_seed(args.seed)
                synthetic = DATASET.PointwisePedDataset()
                synthetic.load_data(yamls['pretrain'])
                synthetic.build_dataset(args)
                np.random.seed(args.seed)
                state = np.random.get_state()
                out['perm_head'] = np.random.permutation(len(synthetic.train_data))[:512]
                np.random.set_state(state)
                loaders = LOADER.data_loader(synthetic.train_data, args.batch_size, args.seed, shuffle=args.shuffle,
                                             drop_last=True)
                torch.manual_seed(args.seed)
                sim = SIM.BaseSimulator(args)
                for k, v in sim.model.state_dict().items():
                    out[f'init/{k}'] = v.clone()
                print('@@ stage pretrain')
                sim.train(loaders, synthetic.valid_data)
                best_pre = torch.load(f'../saved_model/{args.exp_name}_{args.model_name_suffix}')
                real = DATASET.TimeIndexedPedDataset2()
                real.load_data(yamls['finetune'])
                real.build_dataset(args)
                ft_loaders = LOADER.data_loader(real.train_data, args.ft_batch_size, args.seed, shuffle=args.shuffle,
                                                drop_last=True)
                print('@@ stage finetune')
                sim.finetune(ft_loaders, real.valid_data, real.test_data)
                best_ft = torch.load(f'../saved_model/{args.exp_name}_{args.model_name_suffix}_finetuned')
            text = log.getvalue()
            pre, ft = text.split('@@ stage finetune')
            pre = pre.split('@@ stage pretrain')[1]

            def grab(txt, pattern):
                return np.array([[float(x) for x in (m if isinstance(m, tuple) else (m,))]
                                 for m in re.findall(pattern, txt)], np.float64)
            out['pre/train'] = grab(pre, r'Training loss:' + float_re + r', mse:' + float_re)            # (epochs, 2)
            out['pre/val'] = grab(pre, r'Validation loss:' + float_re + r', val_mse:' + float_re)
            out['pre/saved_epochs'] = grab(pre, r'Model Saved at epoch (\d+)')
            out['ft/train'] = grab(ft, r'Training loss:' + float_re + r', mse:' + float_re + r', coll_pred:' + float_re +
                                   r', acc_pred:' + float_re + r', coll:' + float_re + r', hard_coll:' + float_re)
            out['ft/train_collisions'] = grab(ft, r'training collision count hard/soft: ' + float_re + r' & ' + float_re)
            out['ft/val'] = grab(ft, r'Validation loss:' + float_re + r', val_mse:' + float_re)  # [before epoch 0, epoch 0, ...]
            out['ft/saved_epochs'] = grab(ft, r'Model Saved at epoch (\d+)')
            out['ft/test'] = grab(ft, r'Test loss:' + float_re + r', test_mse:' + float_re + r', test_mae:' + float_re +
                                  r', test ot:' + float_re + r', test mmd:' + float_re)          # last row = final test
            out['ft/collisions'] = grab(ft, r'test/val collision count hard/soft: ' + float_re + r' & ' + float_re)
            for tag, sd in (('best_pre', best_pre), ('best_ft', best_ft)):
                for k, v in sd.items():
                    if k.startswith(('ped_encoder.mlp.0', 'ped_encoder.mlp.4', 'obs_encoder.mlp.2', 'ped_predictor',
                                     'obs_decoder.mlp.0.bias', 'ped_collision_predictor.mlp.2')):
                        out[f'{tag}/{k}'] = v.clone()
                out[f'{tag}/l2'] = np.float64(sum(float((v.double() ** 2).sum()) for v in sd.values()) ** 0.5)
            # the complete best fine-tuned weights: lets the GPU test roll the test clip with EXACTLY the reference's
            # network, separating rollout parity from the (chaotic) accumulation of training differences
            for k, v in best_ft.items():
                out[f'best_ft_full/{k}'] = v.clone()
            # final test rollout with the best fine-tuned weights: positions (short horizon) + FDE of the same masks
            sim.model.load_state_dict(best_ft)
            sim.model.eval()
            d = real.test_data[0]
            with torch.no_grad():
                pred = sim.get_multiple_rollouts(d, t_start=args.skip_frames, load_model=False)
                mask = d.mask_p_pred.long()
                p_pred = sim.post_process(d, pred.position.clone(), pred.mask_p, mask)
            labels = d.labels[..., :2]
            out['test/fde'] = np.float64(fde_with_time_mask(p_pred, labels, mask, reduction='mean'))
            out['test/mae'] = np.float64(METRIC.mae_with_time_mask(p_pred, labels, mask, reduction='mean'))
            out['test/rollout_head'] = pred.position[:args.skip_frames + 40].clone()
            out['test/mask_head'] = pred.mask_p[:args.skip_frames + 40].clone()
            err = torch.norm(torch.nan_to_num(p_pred - labels), dim=-1) * (mask == 1)
            out['test/mae_per_frame'] = (err.sum(-1) / (mask == 1).sum(-1).clamp(min=1)).clone()      # (T,)
            out['test/final_position'] = p_pred[-1].clone()
            out['log'] = np.array(text)
            save('mainflow_' + case, **out)
        # the clips the flow reads travel as data fixtures (inputs), next to the other clips under tests/golden/data
        dst = os.path.join(HERE, 'data')
        for case in cases:
            for stage in ('pretrain', 'finetune'):
                for files in MAINFLOW_CASES[case][stage].values():
                    for f in files:
                        if not os.path.exists(os.path.join(dst, os.path.basename(f))):
                            shutil.copy(os.path.join(REF, f), dst)
                ypath = os.path.join(dst, f'mainflow_{case}_{stage}.yaml')
                yaml.safe_dump({k: [os.path.basename(f) for f in v] for k, v in MAINFLOW_CASES[case][stage].items()},
                               open(ypath, 'w'))
    finally:
        os.chdir(cwd)


def gen_mainflow_pointwise(cases=('gc', 'ucy'),
                           init_batches={'gc': (0, 19, 38, 57, 76, 99), 'ucy': (0, 19, 38, 57, 76, 95, 114, 130)},
                           traj_batches={'gc': (99,), 'ucy': (100, 120, 130)}):
    """Single pointwise pre-training steps of the cfg5 flow (src/models/simulators.py:327-360: MSE + message regulariser
    + BCE of the bottleneck collision head), by the reference's own classes:
      b{i}/...     batch i of the first epoch evaluated at the INITIAL weights (= mainflow_*.npz init/): inputs, the three
                   loss terms, the predicted accelerations, every parameter gradient;
      traj{i}/...  batch i along the reference's OWN first-epoch trajectory: the weights it holds when it reaches that batch
                   (captured in front of its optimizer.step), the gradients it computed there, inputs, losses, predictions
                   -- for UCY these sit either side of the batch (~110) where two float32 implementations drift apart."""
    import yaml
    import torch.nn.functional as F
    import data.dataset as DATASET
    import models.simulators as SIM
    import utils.data_loader as LOADER
    scratch = '/tmp/piml_ref_mainflow'
    os.makedirs(os.path.join(scratch, 'src'), exist_ok=True)
    cwd = os.getcwd()
    os.chdir(os.path.join(scratch, 'src'))

    class Done(Exception):
        pass
    try:
        for case in cases:
            cfg = MAINFLOW_CASES[case]
            args = mainflow_args(case, epochs=1)
            path = os.path.join(scratch, f'{case}_pretrain.yaml')
            yaml.safe_dump({k: [os.path.join(REF, f) for f in v] for k, v in cfg['pretrain'].items()}, open(path, 'w'))
            np.random.seed(args.seed)
            torch.manual_seed(args.seed)
            synthetic = DATASET.PointwisePedDataset()
            synthetic.load_data(path)
            synthetic.build_dataset(args)
            np.random.seed(args.seed)
            loaders = LOADER.data_loader(synthetic.train_data, args.batch_size, args.seed, shuffle=args.shuffle, drop_last=True)
            torch.manual_seed(args.seed)
            sim = SIM.BaseSimulator(args)
            committed = np.load(os.path.join(HERE, f'mainflow_{case}.npz'))
            for k, v in sim.model.state_dict().items():       # the same initial weights as the whole-flow fixture
                assert np.array_equal(v.numpy(), committed[f'init/{k}']), k
            out = {}

            def terms(batch):
                ped, obs, selff, labels = batch
                pred = sim.model(ped, obs, selff)
                mse = F.mse_loss(pred[0], labels[:, 4:6], reduction='sum')
                reg = sim.l1_reg_loss(pred[1], args.reg_weight, 'sum')
                bce = F.binary_cross_entropy(pred[-1], labels[:, 6:], reduction='sum')
                return pred, mse, reg, bce

            def record(tag, batch, pred, mse, reg, bce):
                ped, obs, selff, labels = batch
                out[f'{tag}/ped'], out[f'{tag}/obs'], out[f'{tag}/selff'], out[f'{tag}/labels'] = ped, obs, selff, labels
                out[f'{tag}/losses'] = np.array([float(mse), float(reg), float(bce)], np.float64)
                out[f'{tag}/acc'] = pred[0].detach().clone()
            sim.model.train()
            for bi in init_batches[case]:
                sim.model.zero_grad(set_to_none=True)
                pred, mse, reg, bce = terms(loaders[bi])
                (mse + reg + bce).backward()
                record(f'b{bi}', loaders[bi], pred, mse, reg, bce)
                for k, p_ in sim.model.named_parameters():
                    if p_.grad is not None:
                        out[f'b{bi}/grad/{k}'] = p_.grad.clone()
            sim.model.zero_grad(set_to_none=True)
            # the reference's own loop (sim.train), interrupted in front of the optimizer step of the wanted batches
            want = sorted(traj_batches.get(case, ()))
            count = [0]
            real_step = sim.optimizer.step

            def step(*a, **kw):
                b = count[0]
                if b in want:
                    for k, p_ in sim.model.named_parameters():
                        out[f'traj{b}/weights/{k}'] = p_.detach().clone()
                        if p_.grad is not None:
                            out[f'traj{b}/grad/{k}'] = p_.grad.clone()
                    with torch.no_grad():
                        record(f'traj{b}', loaders[b], *terms(loaders[b]))
                    if b == want[-1]:
                        raise Done
                count[0] += 1
                return real_step(*a, **kw)
            sim.optimizer.step = step
            try:
                if want:
                    sim.train(loaders, synthetic.valid_data)
            except Done:
                pass
            save('mainflow_pointwise_' + case, **out)
    finally:
        os.chdir(cwd)


def gen_mainflow_spread(case='ucy'):
    """The REFERENCE against ITSELF: the same train -> finetune -> test sequence as gen_mainflow (same seed, same weights, same
    batches), run again under other -- equally valid -- float32 summation orders of the reference's own CPU kernels:
      threads1    torch.set_num_threads(1)   (the GEMM / reduction blocking follows the thread count)
      threads4    torch.set_num_threads(4)
      nomkldnn    torch.backends.mkldnn disabled (nn.Linear through the native BLAS path)
    and records what moves: the final test row [loss, mse, mae, ot, mmd], the collision counts, the per-epoch fine-tuning
    losses and the best fine-tuned weights' L2 norm + the tensors mainflow_<case>.npz keeps.  The committed fixture of
    gen_mainflow is the 8-thread run.  tests/test_main_gpu.py bounds this package's distance to the reference by the
    reference's own spread (mainflow_<case>_spread.npz)."""
    import contextlib
    import io
    import re
    import yaml
    import data.dataset as DATASET
    import models.simulators as SIM
    import utils.data_loader as LOADER
    scratch = '/tmp/piml_ref_mainflow_spread'
    os.makedirs(os.path.join(scratch, 'src'), exist_ok=True)
    cwd = os.getcwd()
    os.chdir(os.path.join(scratch, 'src'))
    float_re = r'([-+0-9.eE]+|nan|inf)'
    cfg = MAINFLOW_CASES[case]
    out = {}
    try:
        for tag in ('threads8', 'threads1', 'threads4', 'nomkldnn'):
            torch.set_num_threads({'threads1': 1, 'threads4': 4}.get(tag, 8))
            mk = torch.backends.mkldnn.flags(enabled=tag != 'nomkldnn')
            args = mainflow_args(case, exp_name=f'spread_{case}_{tag}')
            yamls = {}
            for stage in ('pretrain', 'finetune'):
                path = os.path.join(scratch, f'{case}_{stage}.yaml')
                yaml.safe_dump({k: [os.path.join(REF, f) for f in v] for k, v in cfg[stage].items()}, open(path, 'w'))
                yamls[stage] = path
            log = io.StringIO()
            with mk, contextlib.redirect_stdout(log):
                np.random.seed(args.seed)
                torch.manual_seed(args.seed)
                synthetic = DATASET.PointwisePedDataset()
                synthetic.load_data(yamls['pretrain'])
                synthetic.build_dataset(args)
                np.random.seed(args.seed)
                loaders = LOADER.data_loader(synthetic.train_data, args.batch_size, args.seed, shuffle=args.shuffle, drop_last=True)
                torch.manual_seed(args.seed)
                sim = SIM.BaseSimulator(args)
                sim.train(loaders, synthetic.valid_data)
                real = DATASET.TimeIndexedPedDataset2()
                real.load_data(yamls['finetune'])
                real.build_dataset(args)
                ft_loaders = LOADER.data_loader(real.train_data, args.ft_batch_size, args.seed, shuffle=args.shuffle, drop_last=True)
                print('@@ stage finetune')
                sim.finetune(ft_loaders, real.valid_data, real.test_data)
                best_ft = torch.load(f'../saved_model/{args.exp_name}_{args.model_name_suffix}_finetuned')
            text = log.getvalue()
            pre, ft = text.split('@@ stage finetune')

            def grab(txt, pattern):
                return np.array([[float(x) for x in (m if isinstance(m, tuple) else (m,))] for m in re.findall(pattern, txt)], np.float64)
            out[f'{tag}/pre_train'] = grab(pre, r'Training loss:' + float_re + r', mse:' + float_re)
            out[f'{tag}/pre_val'] = grab(pre, r'Validation loss:' + float_re + r', val_mse:' + float_re)
            out[f'{tag}/ft_train'] = grab(ft, r'Training loss:' + float_re + r', mse:' + float_re + r', coll_pred:' + float_re +
                                          r', acc_pred:' + float_re + r', coll:' + float_re + r', hard_coll:' + float_re)
            out[f'{tag}/ft_test'] = grab(ft, r'Test loss:' + float_re + r', test_mse:' + float_re + r', test_mae:' + float_re +
                                         r', test ot:' + float_re + r', test mmd:' + float_re)
            out[f'{tag}/collisions'] = grab(ft, r'test/val collision count hard/soft: ' + float_re + r' & ' + float_re)
            out[f'{tag}/l2'] = np.float64(sum(float((v.double() ** 2).sum()) for v in best_ft.values()) ** 0.5)
            for k, v in best_ft.items():
                if k.startswith(('ped_encoder.mlp.0', 'ped_encoder.mlp.2', 'ped_encoder.mlp.4', 'obs_encoder.mlp.2', 'ped_predictor')):
                    out[f'{tag}/w/{k}'] = v.clone()
            print(tag, 'final test row', out[f'{tag}/ft_test'][-1], 'collisions', out[f'{tag}/collisions'][-1], flush=True)
        save(f'mainflow_{case}_spread', **out)
    finally:
        torch.set_num_threads(8)
        os.chdir(cwd)


GENS = dict(mainflow_spread=gen_mainflow_spread, mainflow_pointwise=gen_mainflow_pointwise, mainflow=gen_mainflow, relfeat=gen_relfeat, collision=gen_collision, mlapm=gen_mlapm, calcacc=gen_calcacc, model=gen_model, rollout=gen_rollout, dataset=gen_dataset, metrics=gen_metrics, rollout_more=gen_rollout_more, rollout_flags=gen_rollout_flags, model_polar=gen_model_polar)

if __name__ == '__main__':
    names = sys.argv[1:] or list(GENS)
    for n in names:
        GENS[n]()

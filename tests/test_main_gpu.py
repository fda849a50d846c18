"""End-to-end on the GPU (BASELINE.json configs[0] / configs[4]): the reference's `main.py` flow -- build features
from clips, pre-train pointwise, fine-tune through differentiable rollouts, evaluate rollouts (MSE / MAE = ADE / FDE /
OT / MMD / collisions) -- against numbers captured from the REFERENCE running the same flow on the CPU
(tests/golden/make_golden.py::gen_mainflow -> mainflow_gc.npz / mainflow_ucy.npz; flag sets of the shipped GC and
UCY experiments, dropout 0, 2 + 2 epochs, same initial weights, same batches).

What can and cannot agree: the per-batch arithmetic is pinned to 1e-5 by the operator tests; here errors COMPOUND --
through ~200 Adam updates and through 600-700-frame closed-loop rollouts in which a neighbour entering or leaving the
4 m / view-cone set is a discrete event (hard part ii).  So: training losses and short-horizon
rollouts are held to tight bounds, long-horizon rollout metrics to the stated statistical bounds, and every
measured error is printed."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu
DATA = os.path.join(GOLDEN, 'data')


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-12)))


def test_main_default_flow_pretrain_finetune_rollout():
    from piml_amd import main as MAIN
    history, results = MAIN.main(['-f', '--device', 'cuda:0', '--epochs', '2', '--dataset_name', 'gc1560',
                                  '--dropout', '0.0', '--valid_steps', '5', '--ft_batch_size', '4'])
    assert len(history) >= 3 and all(math.isfinite(h['loss']) for h in history)
    # pointwise pre-training on the toy clip decreases its loss from epoch 0 to 1
    assert history[1]['loss'] < history[0]['loss']
    assert len(results) == 1 and all(math.isfinite(x) for x in results[0])


CASES = {
    # flags of src/configs/exp_configs/piml-gcdata.yaml / piml-ucydata.yaml as make_golden.mainflow_args sets them
    'gc': ['--dataset_name', 'gc2344', '--collision_loss_weight', '200', '--hard_collision_penalty', '2',
           '--val_coll_weight', '30'],
    'ucy': ['--dataset_name', 'ucy', '--collision_loss_weight', '40', '--hard_collision_penalty', '1',
            '--val_coll_weight', '10'],
}
COMMON = ['-f', '--device', 'cuda:0', '--model', 'pinnsf_bm', '--dropout', '0.0', '--learning_rate', '2e-4',
          '--finetune_lr_decay', '0.02', '--batch_size', '128', '--ft_batch_size', '32', '--weight_decay', '1e-6',
          '--valid_steps', '10', '--time_decay', '0.9', '--epochs', '2', '--collision_loss_version', 'v2',
          '--collision_pred_weight', '5e-2', '--reg_weight', '1e-2', '--teacher_weight', '0',
          '--true_label_weight', '0', '--collision_focus_weight', '1', '--patience', '25', '--ft_patience', '5',
          '--seed', '666']

# Bounds (relative unless a name ends in _m = metres), each a small multiple of what is measured on the MI355X
# (DESIGN.md section 2 lists the measured values).  Two groups:
#   * `ref_*`: OUR rollout with EXACTLY the reference's fine-tuned weights -- rollout parity proper, tight (1e-6 m);
#   * the rest: the whole flow with our own training.  GC: negligible (1e-6 .. 3e-5).  UCY has a DISCONTINUITY: two
#     float32 implementations that agree to 1e-5 on every single batch (tools/debug_ucy_batches.py: the fused kernels
#     against this package's library-GEMM path on the same weights, all 131 batches <= 4.5e-5) stay within 2.5e-6 of each
#     other in weight space for ~110 batches and then jump apart by 2.5e-3 within ten batches
#     (tools/debug_ucy_diverge.py): a hidden unit that was dead so far gets a pre-activation within an ulp of zero, one
#     implementation rounds it to +0, the other to a tiny positive value, and Adam turns the first non-zero gradient of
#     the unit's 128 weights -- whatever its size -- into a full lr-sized step.  The library-GEMM path happens to round
#     like the reference's CPU GEMM and follows its trajectory to 1e-6 (PIML_FUSED_ENCODER=0 PIML_FUSED_NETWORK=0
#     PIML_FUSED_ROW_DECODER=0 PIML_FUSED_KSUM_TAIL=0: every line below <= 1e-5; even summing the neighbour axis in a
#     different order than torch.sum moves the final metrics by 3e-2); the fused kernels (bias as the accumulator's initial value, MFMA
#     summation order) do not.  `spread_*` prints the distance between the two paths on every run.  Three builds of
#     round 2 (different MFMA / VALU summation orders) gave for (pre_val, weights, worst metric, collisions):
#     (1.2e-3, 1.6e-3, 1.2e-3, 7e-3), (2.5e-3, 3e-3, 4e-3, 7e-3), (3.1e-3, 4.1e-3, 1.0e-2, 2.0e-2); round 3 (split products the
#     default): (8.7e-4, 2.2e-3, 1.2e-3, 6.7e-3), ft_train 1.6e-2.  The bounds are ~3x the largest seen, ft_train / weights
#     tightened in round 3.  The mechanism itself is asserted in tests/test_ucy_gpu.py (gradients on the same weights agree
#     on all 131 batches to 4.4e-5; the trajectories separate at ONE hidden unit), the per-step arithmetic against the
#     reference in the single-step test below (incl. the reference's own weights at batches 100 / 120 / 130).
#     `python -m piml_amd.main --library_gemm 1` trains on the path that reproduces the reference's numbers to 1e-6.
TOL = {
    # round 6: <= 3x this build's measured values (gpurun_out/r6_b/gc.log: pre_train 6.5e-8, pre_val 5.6e-7, ft_train 9.8e-7,
    # weights 6.2e-5, val 4.8e-3 -- the second epoch's rollout validation, one collision window apart --, metrics <= 7.8e-6,
    # reference-weights rollout 3.8e-6 m over 10 frames / 9.5e-6 m over 40, its metrics 1.4e-6, MAE per frame 3.6e-6 m)
    'gc': dict(pre_train=2e-7, pre_val=2e-6, ft_train=3e-6, ft_counts=0.0, weights=2e-4, val=1.5e-2, metrics=2.5e-5,
               collisions=0.0, ref_first10_m=1.2e-5, ref_first40_m=3e-5, ref_metrics=5e-6, ref_mae_per_frame_m=1.1e-5,
               ref_collisions=0.0),
    # round 4: <= 3x this build's measured values (metrics 5.4e-4, weights 2.6e-3, ft_train 1.4e-3 .. 1.6e-2 over the builds,
    # collisions 6.7e-3 = 2 of 296), AND the distance is bounded by the reference's own spread below (<= 2x).
    'ucy': dict(pre_train=2e-4, pre_val=2e-3, ft_train=5e-2, ft_counts=3e-2, weights=7e-3, val=None, metrics=4e-3,
                collisions=2e-2, ref_first10_m=1e-4, ref_first40_m=1e-4, ref_metrics=1e-4, ref_mae_per_frame_m=1e-4,
                ref_collisions=0.0),
}


@pytest.mark.parametrize('case', ['gc', 'ucy'])
def test_main_flow_matches_reference_end_to_end(case):
    from piml_amd import main as MAIN
    from piml_amd.functions import metrics as METRIC
    g = golden('mainflow_' + case)
    tol = TOL[case]
    init = {k[5:]: torch.tensor(g[k]) for k in g.files if k.startswith('init/')}
    argv = COMMON + CASES[case] + ['--data_config', os.path.join(DATA, f'mainflow_{case}_pretrain.yaml'),
                                   '--ft_data_config', os.path.join(DATA, f'mainflow_{case}_finetune.yaml')]
    MAIN.main(argv, init_state=init)
    run = MAIN.LAST_RUN
    sim = run['simulator']
    report = {}

    # the pointwise batches are the reference's (one global-RNG permutation)
    np.random.seed(666)
    assert np.array_equal(np.random.permutation(run['n_train'])[:512], g['perm_head'])

    # ---- pre-training: per-epoch training loss / mse and validation loss ----
    pre = run['pretrain_history']
    got = np.array([[h['loss'], h['mse']] for h in pre])
    report['pre_train'] = rel(got, g['pre/train'])
    report['pre_val'] = rel([h['val_loss'] for h in pre], g['pre/val'][:, 0])
    assert [i for i, h in enumerate(pre) if h.get('saved')] == [int(x) for x in g['pre/saved_epochs'][:, 0]]

    # ---- fine-tuning: per-epoch training terms, collision bookkeeping, model selection ----
    ft = run['finetune_history']
    got = np.array([[h['loss'], h['mse'], h['collision_pred'], h['acc_pred'], h['collision'], h['hard_collision']]
                    for h in ft])
    want = g['ft/train']
    report['ft_train'] = rel(got[:, [0, 1, 2, 3]], want[:, [0, 1, 2, 3]])
    # the collision-focus terms are sums over agents that collided somewhere in a window (discrete membership)
    report['ft_train_collision_terms'] = rel(got[:, 4:] + 1e-3, want[:, 4:] + 1e-3)
    report['ft_train_collision_counts'] = rel(np.array([h['train_collisions'] for h in ft], np.float64) + 1.0,
                                              g['ft/train_collisions'] + 1.0)
    val_got = np.array([sim.initial_val_loss] + [h['val_loss'] for h in ft])
    report['val'] = rel(val_got, g['ft/val'][:, 0])
    saved_got = [i for i, h in enumerate(ft) if h.get('saved')]
    report['saved_epochs'] = (saved_got, [int(x) for x in g['ft/saved_epochs'][:, 0]])

    # ---- weights after each stage ----
    for tag, sd in (('best_pre', sim._checkpoints[False]), ('best_ft', sim._checkpoints[True])):
        worst = 0.0
        for k in g.files:
            if k.startswith(tag + '/') and not k.endswith('/l2'):
                w = g[k]
                worst = max(worst, float(np.abs(sd[k[len(tag) + 1:]].cpu().numpy() - w).max() / max(np.abs(w).max(), 1e-12)))
        report['weights_' + tag] = worst

    # ---- final test (the last printed "Test loss" line): MSE / MAE (= ADE) / OT / MMD, collisions, FDE ----
    ev = sim.finetune_test_result
    last = g['ft/test'][-1]
    report['test_mse'] = rel(ev[1], last[1])
    report['mae'] = rel(ev[2], last[2])
    report['ot'] = rel(ev[3], last[3])
    report['mmd'] = rel(ev[4], last[4])
    report['fde'] = rel(sim.last_eval['fde'], g['test/fde'])
    report['collisions'] = rel(np.array([sim.last_eval['hard_collisions'], sim.last_eval['collisions']]) + 1.0,
                               g['ft/collisions'][-1] + 1.0)

    # ---- rollout parity proper: OUR rollout with EXACTLY the reference's best fine-tuned weights (the training
    # differences above do not enter): short horizon in metres, then the whole-horizon metrics ----
    d = run['finetune_data'].test_data[0]
    skip = run['args'].skip_frames
    ref_sd = {k[len('best_ft_full/'):]: torch.tensor(g[k]) for k in g.files if k.startswith('best_ft_full/')}
    own_sd = {k: v.clone() for k, v in sim.model.state_dict().items()}
    sim.model.load_state_dict(ref_sd)
    with torch.no_grad():
        sim.model.eval()
        pred = sim.get_multiple_rollouts(d, t_start=skip, load_model=False)
        mask = d.mask_p_pred.long()
        p_post = sim.post_process(d, pred.position.clone(), pred.mask_p, mask)
        labels = d.labels[..., :2]
        rw = dict(mae=METRIC.mae_with_time_mask(p_post, labels, mask, reduction='mean'),
                  fde=METRIC.fde_with_time_mask(p_post, labels, mask, reduction='mean'),
                  coll=METRIC.collision_count(pred.position[skip:], run['args'].collision_threshold, reduction='sum'),
                  hard=METRIC.collision_count(pred.position[skip:], run['args'].collision_threshold / 2, reduction='sum'))
        err = torch.norm(torch.nan_to_num(p_post - labels), dim=-1) * (mask == 1)
        mae_t = (err.sum(-1) / (mask == 1).sum(-1).clamp(min=1)).cpu().numpy()
    sim.model.load_state_dict(own_sd)
    head = torch.nan_to_num(pred.position[:skip + 40]).cpu().numpy()
    want_head = np.nan_to_num(g['test/rollout_head'])
    herr = np.abs(head - want_head).max(axis=(1, 2))
    report['refweights_rollout_first10_m'] = float(herr[skip:skip + 10].max())
    report['refweights_rollout_first40_m'] = float(herr.max())
    report['refweights_mask_head_equal'] = bool(np.array_equal(pred.mask_p[:skip + 40].cpu().numpy() > 0, g['test/mask_head'] > 0))
    report['refweights_mae'] = rel(rw['mae'], g['test/mae'])
    report['refweights_fde'] = rel(rw['fde'], g['test/fde'])
    report['refweights_collisions'] = rel(np.array([rw['hard'], rw['coll']]) + 1.0, g['ft/collisions'][-1] + 1.0)
    dm = np.abs(mae_t - g['test/mae_per_frame'])
    report['refweights_mae_per_frame_first100_m'] = float(dm[:skip + 100].max())
    report['refweights_mae_per_frame_all_m'] = float(dm.max())

    if case == 'ucy':
        # the same flow on this package's OTHER float32 path (library GEMMs instead of the fused MFMA kernels): how far
        # two correct implementations drift apart on this chaotic configuration
        import piml_amd.models.model as MODEL
        first = dict(mse=ev[1], mae=ev[2], ot=ev[3], mmd=ev[4], fde=sim.last_eval['fde'],
                     val=val_got.copy(), pre_val=np.array([h['val_loss'] for h in pre]))
        old_flags = (MODEL.FUSED_ENCODER, MODEL.FUSED_NETWORK, MODEL.FUSED_ROW_DECODER, MODEL.FUSED_KSUM_TAIL)
        try:
            MAIN.main(argv + ['--library_gemm', '1'], init_state=init)
            assert not (MODEL.FUSED_ENCODER or MODEL.FUSED_NETWORK or MODEL.FUSED_ROW_DECODER or MODEL.FUSED_KSUM_TAIL)
        finally:
            MODEL.FUSED_ENCODER, MODEL.FUSED_NETWORK, MODEL.FUSED_ROW_DECODER, MODEL.FUSED_KSUM_TAIL = old_flags
        sim2 = MAIN.LAST_RUN['simulator']
        ev2 = sim2.finetune_test_result
        report['spread_metrics(fused vs library path)'] = max(rel(first['mse'], ev2[1]), rel(first['mae'], ev2[2]),
                                                              rel(first['ot'], ev2[3]), rel(first['mmd'], ev2[4]),
                                                              rel(first['fde'], sim2.last_eval['fde']))
        report['spread_pre_val'] = rel(first['pre_val'], [h['val_loss'] for h in MAIN.LAST_RUN['pretrain_history']])
        report['library_path_vs_reference_metrics'] = max(rel(ev2[1], last[1]), rel(ev2[2], last[2]), rel(ev2[3], last[3]),
                                                          rel(ev2[4], last[4]))
    if case == 'ucy':
        # The REFERENCE against ITSELF (tests/golden/make_golden.py::gen_mainflow_spread -> mainflow_ucy_spread.npz): the same flow,
        # same seed, same batches, under other float32 summation orders of its own CPU kernels (1 / 4 / 8 torch threads, oneDNN
        # off).  1 thread and oneDNN-off reproduce the 8-thread fixture to 1e-9; FOUR threads move the final metrics by 4.4e-4,
        # the second epoch's fine-tuning loss by 1.6e-2, the best weights by 1.6e-3 and the soft collision count from 296 to 298:
        # the configuration is chaotic in the reference itself, by the amounts this package's fused path differs from it.
        sp = golden('mainflow_ucy_spread')
        tags = [t for t in ('threads1', 'threads4', 'nomkldnn')]
        base = sp['threads8/ft_test'][-1]
        assert np.allclose(base, last, rtol=1e-12), 'the spread fixture was generated from another reference run'
        ref_metrics = max(rel(sp[f'{t}/ft_test'][-1][1:5], base[1:5]) for t in tags)
        ref_ft_train = max(rel(sp[f'{t}/ft_train'][:, :4], sp['threads8/ft_train'][:, :4]) for t in tags)
        ref_coll = max(rel(sp[f'{t}/collisions'][-1] + 1.0, sp['threads8/collisions'][-1] + 1.0) for t in tags)
        ref_w = 0.0
        for k in sp.files:
            if k.startswith('threads8/w/'):
                w = sp[k]
                ref_w = max(ref_w, max(float(np.abs(sp[k.replace('threads8', t)] - w).max() / max(np.abs(w).max(), 1e-12)) for t in tags))
        report['reference_self_spread(metrics, ft_train, weights, collisions)'] = (ref_metrics, ref_ft_train, ref_w, ref_coll)
        own_metrics = max(report['test_mse'], report['mae'], report['ot'], report['mmd'])
        report['distance / reference self-spread (metrics, ft_train, weights, collisions)'] = (
            own_metrics / ref_metrics, report['ft_train'] / ref_ft_train, report['weights_best_ft'] / ref_w, report['collisions'] / ref_coll)
    print(f'\n[cfg5 {case}] measured deviations from the reference (relative unless noted):')
    for k, v in report.items():
        print(f'    {k:32s} {v}')
    print(f'    reference: ft val {g["ft/val"][:, 0]}, test mse/mae/ot/mmd {last[1:]}, fde {g["test/fde"]}, '
          f'collisions hard/soft {g["ft/collisions"][-1]}')
    print(f'    here:      ft val {val_got}, test mse/mae/ot/mmd {ev[1:]}, fde {sim.last_eval["fde"]}, collisions '
          f'hard/soft {[sim.last_eval["hard_collisions"], sim.last_eval["collisions"]]}')

    # rollout parity with the reference's weights
    assert report['refweights_mask_head_equal']
    assert report['refweights_rollout_first10_m'] <= tol['ref_first10_m']
    assert report['refweights_rollout_first40_m'] <= tol['ref_first40_m']
    assert report['refweights_mae_per_frame_all_m'] <= tol['ref_mae_per_frame_m']
    assert report['refweights_mae'] <= tol['ref_metrics'] and report['refweights_fde'] <= tol['ref_metrics']
    assert report['refweights_collisions'] <= tol['ref_collisions']
    # the whole flow
    assert report['pre_train'] <= tol['pre_train'] and report['pre_val'] <= tol['pre_val']
    assert report['ft_train'] <= tol['ft_train'] and report['ft_train_collision_counts'] <= tol['ft_counts']
    assert report['weights_best_pre'] <= tol['weights'] and report['weights_best_ft'] <= tol['weights']
    assert report['saved_epochs'][0] == report['saved_epochs'][1]          # same model selection
    if case == 'ucy':
        # the per-epoch rollout validation of the chaotic UCY flow: bounded like the other quantities below by twice what the
        # reference's OWN thread-count spread does to the same kind of number (the per-epoch test mse of the spread fixture: 2.8e-3
        # between 4 and 8 torch threads).  Measured here 7e-4 .. 3.0e-3 depending on the summation order of a weight-gradient
        # kernel (round 6: the collision head's backward went from a four-wave serial sum to disjoint blocks and moved it from
        # 8e-4 to 3.0e-3 -- with the GC flow, which is not chaotic, unchanged at 5e-3 of ITS tolerance).
        ref_val = max(rel(sp[f'{t}/ft_test'][:, 0], sp['threads8/ft_test'][:, 0]) for t in tags)
        report['val / reference self-spread of the per-epoch test mse'] = report['val'] / ref_val
        assert ref_val >= 1e-3 and report['val'] <= 2.0 * ref_val, report
    else:
        assert report['val'] <= tol['val']
    for k in ('test_mse', 'mae', 'fde', 'ot', 'mmd'):
        assert report[k] <= tol['metrics'], k
    assert report['collisions'] <= tol['collisions']
    if case == 'ucy':       # this package's library-GEMM path follows the reference's trajectory itself (see TOL)
        assert report['library_path_vs_reference_metrics'] <= 1e-4
        # ... and the fused path is no further from the reference than twice the reference is from itself
        assert ref_metrics >= 1e-4, 'the reference reproduces itself: the divergence would be this package\'s'
        for d in report['distance / reference self-spread (metrics, ft_train, weights, collisions)']:
            assert d <= 2.0, report


@pytest.mark.parametrize('case', ['gc', 'ucy'])
def test_pointwise_training_step_matches_reference(case):
    """Single pointwise pre-training steps (src/models/simulators.py:327-356: MSE + message regulariser + BCE of the
    bottleneck collision head) of the cfg5 flow against the reference's (tests/golden/mainflow_pointwise_*.npz,
    make_golden.py::gen_mainflow_pointwise): loss terms, predictions and every parameter gradient --
      b*     6 / 8 batches spread over the first epoch at the reference's INITIAL weights,
      traj*  batches along the reference's OWN trajectory, at the weights it holds when it gets there (UCY: 100, 120, 130,
             either side of the batch where two float32 implementations drift apart; GC: the last one) --
    the per-step bar under the compounded epoch numbers of the test above."""
    from piml_amd import main as MAIN
    from piml_amd.models import simulators as SIM
    import torch.nn.functional as F
    g = golden('mainflow_' + case)
    pw = golden('mainflow_pointwise_' + case)
    args = MAIN.get_args(COMMON + CASES[case])
    args.ped_feature_dim, args.obs_feature_dim, args.self_feature_dim = 6, 6, 7
    sim = SIM.BaseSimulator(args)
    init = {k[5:]: torch.tensor(g[k]) for k in g.files if k.startswith('init/')}
    sim.model.train()
    tags = sorted({k.split('/')[0] for k in pw.files}, key=lambda t: (t.startswith('traj'), int(t.lstrip('btraj'))))
    assert len(tags) >= 7
    worst = {}
    for tag in tags:
        sd = dict(init)
        if tag.startswith('traj'):
            sd.update({k[len(tag) + 9:]: torch.tensor(pw[k]) for k in pw.files if k.startswith(tag + '/weights/')})
        sim.model.load_state_dict(sd)
        ped, obs, selff, labels = [torch.tensor(pw[f'{tag}/{k}'], device='cuda:0') for k in ('ped', 'obs', 'selff', 'labels')]
        sim.model.zero_grad(set_to_none=True)
        pred = sim.model(ped, obs, selff)
        mse = F.mse_loss(pred[0], labels[:, 4:6], reduction='sum')
        reg = sim.l1_reg_loss(pred[1], args.reg_weight, 'sum')
        bce = F.binary_cross_entropy(pred[-1], labels[:, 6:], reduction='sum')
        (mse + reg + bce).backward()
        w = {'losses': rel([float(mse), float(reg), float(bce)], pw[f'{tag}/losses']),
             'acc': float(np.abs(pred[0].detach().cpu().numpy() - pw[f'{tag}/acc']).max() / np.abs(pw[f'{tag}/acc']).max()),
             'grad': 0.0}
        for k, p in sim.model.named_parameters():
            key = f'{tag}/grad/{k}'
            if key in pw.files:
                ref = pw[key]
                e = float(np.abs(p.grad.cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-12))
                if e > w['grad']:
                    w['grad'], w['grad_worst'] = e, k
        print(f'[cfg5 {case}] pointwise step {tag} vs reference: ' + ', '.join(f'{k} {v}' for k, v in w.items()))
        for k in ('losses', 'acc', 'grad'):
            worst[k] = max(worst.get(k, 0.0), w[k])
    print(f'[cfg5 {case}] worst over {len(tags)} steps: {worst}')
    assert worst['losses'] <= 1e-5 and worst['acc'] <= 1e-5 and worst['grad'] <= 1e-4

"""End-to-end plumbing on the GPU (BASELINE.json configs[0] / configs[4] shaped): the reference's
`main.py` flow -- build features from clips, pre-train pointwise, fine-tune through differentiable
rollouts, roll a real GC clip out and count collisions."""
import math

import pytest

pytestmark = pytest.mark.gpu


def test_main_default_flow_pretrain_finetune_rollout():
    from piml_amd import main as MAIN
    history, results = MAIN.main(['-f', '--device', 'cuda:0', '--epochs', '2', '--dataset_name', 'gc1560',
                                  '--dropout', '0.0', '--valid_steps', '5', '--ft_batch_size', '4'])
    assert len(history) >= 3 and all(math.isfinite(h['loss']) for h in history)
    # pointwise pre-training on the toy clip decreases its loss from epoch 0 to 1
    assert history[1]['loss'] < history[0]['loss']
    assert len(results) == 1 and all(math.isfinite(x) for x in results[0])

"""CPU: the guard around the HIP-graph workaround (piml_amd/__init__.py).  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 only works when it
is in the environment before the HIP runtime initialises; a process where that cannot be guaranteed must not replay
captured graphs ("HIP-graph hazard")."""
import os
import subprocess
import sys

from conftest import REPO


def test_state_function():
    import piml_amd
    f = piml_amd._graph_env_state
    assert f('0', False) == (True, None) and f('0', True) == (True, None)
    assert f(None, False) == (True, None)                       # import sets it in time
    safe, why = f(None, True)
    assert not safe and 'before piml_amd was imported' in why
    safe, why = f('1', False)
    assert not safe and 'explicitly' in why


def run(code, env_extra):
    env = dict(os.environ, PYTHONPATH=REPO, **env_extra)
    for k in [k for k, v in env_extra.items() if v is None]:
        env.pop(k)
    return subprocess.run([sys.executable, '-W', 'always', '-c', code], capture_output=True, text=True, env=env, timeout=300)


def test_explicit_other_value_disables_graphs_loudly():
    r = run('import piml_amd; print(piml_amd.hip_graphs_safe())', {'DEBUG_CLR_GRAPH_PACKET_CAPTURE': '1'})
    assert r.returncode == 0 and r.stdout.strip() == 'False'
    assert 'HIP-graph capture is DISABLED' in r.stderr
    r = run('import piml_amd; print(piml_amd.hip_graphs_safe())', {'DEBUG_CLR_GRAPH_PACKET_CAPTURE': '1', 'PIML_TRUST_HIP_GRAPHS': '1'})
    assert r.stdout.strip() == 'True'


def test_default_import_sets_the_variable_and_allows_graphs():
    r = run('import os, piml_amd; print(piml_amd.hip_graphs_safe(), os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"])',
            {'DEBUG_CLR_GRAPH_PACKET_CAPTURE': None})
    assert r.returncode == 0 and r.stdout.split() == ['True', '0'] and 'DISABLED' not in r.stderr


def test_simulator_and_bench_ask_the_guard():
    """Every capture site consults hip_graphs_safe()."""
    for rel in ('piml_amd/models/simulators.py', 'piml_amd/models/mlapm.py', 'bench.py'):
        text = open(os.path.join(REPO, rel)).read()
        assert text.count('CUDAGraph()') >= 1 and 'hip_graphs_safe()' in text, rel

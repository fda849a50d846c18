"""`MLAPM`: the closed-form social-force law (reference src/models/mlapm.py), same constructor
and `step` signature, evaluated by the HIP pair kernel with an analytic backward."""
from .. import ops


class MLAPM:
    def __init__(self, **args):
        self.args = args
        if args.get('version') not in ops.MLAPM_VARIANTS:
            raise NotImplementedError(args.get('version'))

    def step(self, position, velocity, desired_speed, destination, dt, radius=0.3):
        """position, velocity, destination: (N, 2); desired_speed: (N, 1).  Returns the new
        velocity `velocity + force * dt` (mlapm.py:10-58).  As in the reference, absent (NaN)
        agents must be filtered out by the caller.  Deviation: version 'UCY' applies the
        one-line `coll.unsqueeze(-1)` fix without which the reference raises for N > 2."""
        a = self.args
        return ops.mlapm_step(position, velocity, desired_speed, destination, dt, radius, version=a['version'],
                              tau=a['tau'], A=a['A'], B=a['B'], C=a.get('C', 0.0), D=a.get('D', 0.0),
                              theta=a.get('theta', 0.0))

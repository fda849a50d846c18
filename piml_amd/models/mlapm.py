"""`MLAPM`: the closed-form social-force law (reference src/models/mlapm.py), same constructor
and `step` signature, evaluated by the HIP pair kernel with an analytic backward."""
from .. import ops


class MLAPM:
    def __init__(self, **args):
        self.args = args
        if args.get('version') not in ops.MLAPM_VARIANTS:
            raise NotImplementedError(args.get('version'))

    def step(self, position, velocity, desired_speed, destination, dt, radius=0.3):
        """position, velocity, destination: (N, 2); desired_speed: (N, 1), (N,) or (N, 2).  Returns the new
        velocity `velocity + force * dt` (mlapm.py:10-58).  As in the reference, absent (NaN)
        agents must be filtered out by the caller.  Deviation: version 'UCY' applies the
        one-line `coll.unsqueeze(-1)` fix without which the reference raises for N > 2."""
        a = self.args
        kw = dict(version=a['version'], tau=a['tau'], A=a['A'], B=a['B'], C=a.get('C', 0.0), D=a.get('D', 0.0),
                  theta=a.get('theta', 0.0))
        N = position.shape[0]
        if desired_speed.dim() == 2 and tuple(desired_speed.shape) == (N, 2):
            # the reference's own driver passes (N, 2) (src/main_mlapm.py:13): `desired_speed * ed` then uses a
            # per-component speed.  The kernel takes the x column; the y component's difference is a linear
            # correction of the desired-force term, (0, (v0y - v0x) * ed_y) / tau * dt (exactly zero for equal columns).
            import torch.nn.functional as F
            base = ops.mlapm_step(position, velocity, desired_speed[:, :1], destination, dt, radius, **kw)
            ed_y = F.normalize(destination - position, dim=-1, p=2)[:, 1]
            corr = (desired_speed[:, 1] - desired_speed[:, 0]) * ed_y * (dt / a['tau'])
            return base + F.pad(corr.unsqueeze(-1), (1, 0))
        return ops.mlapm_step(position, velocity, desired_speed, destination, dt, radius, **kw)

    def rollout(self, position, velocity, desired_speed, destination, dt, radius=0.3, steps=200, use_graph=True, fused=True,
                frames_per_graph=8):
        """The simulation loop of src/main_mlapm.py:18-36 on the device: step, explicit Euler
        `p += v dt`, and agents within `radius` of their destination leave the scene (NaN from the next
        frame on).  The reference compacts the active agents on the host every step; here absent
        agents stay in place as NaN rows (`skip_absent`), so shapes are static.  Returns positions and
        velocities (steps + 1, N, 2).
        fused (default): a frame is ONE launch (`ops.mlapm_rollout_step`: the state is read from the trajectory, the frame
        counter lives on the device) and `frames_per_graph` of them are one captured HIP graph; fused=False: the operator
        sequence (MLAPM.step + torch glue, one captured frame replayed)."""
        import torch
        a = self.args
        N = position.shape[0]
        if fused and position.is_cuda:
            spd = desired_speed
            if spd.dim() == 2 and tuple(spd.shape) == (N, 2) and bool((spd[:, 0] == spd[:, 1]).all()):
                spd = spd[:, :1]                              # the reference's driver passes two equal columns (main_mlapm.py:13)
            if spd.numel() == N:
                return self._rollout_fused(position, velocity, spd, destination, dt, radius, steps, use_graph, frames_per_graph)
        p, v = position.clone(), velocity.clone()
        traj_p = torch.full((steps + 1, N, 2), float('nan'), device=p.device)
        traj_v = torch.full((steps + 1, N, 2), float('nan'), device=p.device)
        traj_p[0], traj_v[0] = p, v
        t = torch.ones(1, dtype=torch.long, device=p.device)
        nan = torch.tensor(float('nan'), device=p.device)

        def one():
            v_new = ops.mlapm_step(p, v, desired_speed, destination, dt, radius, version=a['version'], tau=a['tau'],
                                   A=a['A'], B=a['B'], C=a.get('C', 0.0), D=a.get('D', 0.0),
                                   theta=a.get('theta', 0.0), skip_absent=True)
            p_new = p + v_new * dt
            traj_p.index_copy_(0, t, p_new.unsqueeze(0))
            traj_v.index_copy_(0, t, v_new.unsqueeze(0))
            arrived = (torch.norm(p_new - destination, dim=-1, keepdim=True) < radius)
            p.copy_(torch.where(arrived, nan, p_new))
            v.copy_(torch.where(arrived, nan, v_new))
            t.add_(1)
        done = 0
        with torch.no_grad():
            from .. import hip_graphs_safe
            if use_graph and p.is_cuda and steps > 4 and hip_graphs_safe():
                for _ in range(2):
                    one()
                done = 2
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    one()
                for _ in range(steps - done):
                    g.replay()
                done = steps
            for _ in range(steps - done):
                one()
        return traj_p, traj_v

    def _rollout_fused(self, position, velocity, desired_speed, destination, dt, radius, steps, use_graph, frames_per_graph):
        import torch
        a = self.args
        dev, N = position.device, position.shape[0]
        kw = dict(version=a['version'], tau=a['tau'], A=a['A'], B=a['B'], C=a.get('C', 0.0), D=a.get('D', 0.0),
                  theta=a.get('theta', 0.0))
        with torch.no_grad():
            traj_p = torch.full((steps + 1, N, 2), float('nan'), device=dev)
            traj_v = torch.full((steps + 1, N, 2), float('nan'), device=dev)
            traj_p[0], traj_v[0] = position, velocity
            v0 = desired_speed.reshape(N, 1).float().contiguous()
            dest = destination.float().contiguous()
            t = torch.ones(1, dtype=torch.int64, device=dev)
            fin = torch.zeros(1, dtype=torch.int32, device=dev)

            def one():
                ops.mlapm_rollout_step(traj_p, traj_v, v0, dest, t, fin, dt, radius, **kw)
            from .. import hip_graphs_safe
            done = 0
            per = max(1, int(frames_per_graph))
            if use_graph and steps >= 2 * per + 2 and hip_graphs_safe():
                one()                                         # a real frame, also warms the library up
                done = 1
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(per):
                        one()
                for _ in range((steps - done) // per):       # (a launch past the last frame does nothing)
                    g.replay()
                done += (steps - done) // per * per
            for _ in range(steps - done):
                one()
        return traj_p, traj_v

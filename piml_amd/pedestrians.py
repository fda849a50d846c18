"""`Pedestrians`: the reference's pairwise-geometry operator class (src/data/data.py:343-601),
same method names / arguments / return conventions, backed by the HIP kernels.

Drop-in point: `BaseSimulator(DATA.Pedestrians)` (src/models/simulators.py:25) and
`TimeIndexedPedData(Dataset, Pedestrians)` (src/data/data.py:604) inherit from this class.
"""
from . import ops


class Pedestrians(object):

    def __init__(self):
        super(Pedestrians, self).__init__()

    @staticmethod
    def get_heading_direction(velocity):
        """velocity (*c, t, N, 2) -> unit heading with temporal zero-fill (data.py:350-395)."""
        return ops.heading_direction(velocity)

    def get_relative_features(self, position, velocity, acceleration, destination, obstacles,
                              topk_ped, sight_angle_ped, dist_threshold_ped, topk_obs,
                              sight_angle_obs, dist_threshold_obs):
        """Same contract as data.py:466-512: inputs (*c, t, N, 2), obstacles (M, 2); returns
        (ped_features (*c,t,N,k_p,6), obs_features (*c,t,N,k_o,6), dest_features (*c,t,N,2)).
        Like the reference it zeroes NaNs of `velocity` / `acceleration` IN PLACE first
        (data.py:483-484).  Deviation: with no obstacles the reference returns an empty
        (t, 0) tensor; here obs_features is (*c, t, N, 0, 6)."""
        acceleration.masked_fill_(acceleration.isnan(), 0)
        velocity.masked_fill_(velocity.isnan(), 0)
        num_steps = position.shape[-3]
        heading = None if num_steps == 1 else ops.heading_direction(velocity.detach())
        return ops.relative_features(
            position, velocity, acceleration, destination, obstacles,
            topk_ped, sight_angle_ped, dist_threshold_ped, topk_obs, sight_angle_obs,
            dist_threshold_obs, heading=heading)

    @staticmethod
    def calculate_collision_label(ped_features):
        """(..., k, 6) -> (..., k): collides within 1 s at the current relative velocity
        (data.py:514-535)."""
        return ops.collision_label(ped_features)

    @staticmethod
    def collision_detection(position, threshold, real_position=None):
        """(t,n,2) / (c,n,2) / (c,t,n,2) -> same-rank (..., n, n) 0/1 matrix with the reference's
        "friends" filters (data.py:537-601, quirk Q7).  Callers that only need per-agent counts
        should use `ops.collision_counts`, which never builds the matrix."""
        return ops.collision_detection(position, threshold, real_position)

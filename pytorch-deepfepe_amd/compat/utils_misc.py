"""Mirror of the one helper of deepFEPE/dsac_tools/utils_misc.py that the match-construction step needs."""
import numpy as np


def crop_or_pad_choice(in_num_points, out_num_points, shuffle=False):
    """Indices that crop or pad ``in_num_points`` items to ``out_num_points`` (utils_misc.py:139-161).  Host-side and
    drawn from numpy's global RNG with the reference's call sequence (one permutation when shuffling, one
    ``np.random.choice`` when padding), so a seeded run selects the same correspondences as the reference."""
    choice = np.random.permutation(in_num_points) if shuffle else np.arange(in_num_points)
    assert out_num_points > 0, "out_num_points = %d must be positive int!" % out_num_points
    if in_num_points >= out_num_points:
        return choice[:out_num_points]
    pad = np.random.choice(choice, out_num_points - in_num_points, replace=True)
    return np.concatenate([choice, pad])

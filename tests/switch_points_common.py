"""Shared by test_switch_points_do_not_change_results and its child process: a fixed set of K_EMBED agents
embedded at the front of every batch, the rest of the batch filled from bench.py's generator."""
import hashlib
import numpy as np

N, K_EMBED = 20, 700


def batch(B):
    import bench
    rng = np.random.default_rng(77)
    emb = np.stack([rng.uniform(0, 5, K_EMBED), rng.uniform(-.3, .3, K_EMBED), rng.uniform(-.3, .3, K_EMBED),
                    rng.uniform(.3, 1.5, K_EMBED)], 1)
    X = bench.synthetic_states(0, 0, B)
    X[:K_EMBED] = emb
    return X


def digest(U, st):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(U[:K_EMBED].cpu().numpy()).tobytes())
    h.update(np.ascontiguousarray(st[:K_EMBED].cpu().numpy()).tobytes())
    return h.hexdigest()

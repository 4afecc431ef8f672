"""Catalogues for the tools that are not uniform noise (tools/run_replica.py, run_batched.py, run_half_multi.py: --catalogue clustered)."""


def clustered_catalogue(n, spread, seed=777, clusters=3000, contiguous=False):
    """Rows shaped like the reference's min-max normalised features (Song.h:18-19: danceability, energy, key, loudness,
    mode, speechiness, acousticness, instrumentalness, liveness, valence, tempo, genre_id — DataManager.cpp:286-299):
    cluster centres in [0,1]^12 with key in {0..11}/11, mode in {0,1}, genre in {0..113}/113, Gaussian spread around the
    continuous columns, and 2 % exact duplicates (the same track on several albums).  contiguous: the rows of a cluster
    lie next to each other (a catalogue sorted by genre / artist): a launch-wide cutoff taken from evenly spaced sample
    regions then mostly MISSES the query's own cluster."""
    import torch
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    centres = torch.rand((clusters, 12), device="cuda", generator=g)
    centres[:, 2] = torch.randint(0, 12, (clusters,), device="cuda", generator=g).float() / 11.0
    centres[:, 4] = torch.randint(0, 2, (clusters,), device="cuda", generator=g).float()
    centres[:, 7] = centres[:, 7] ** 4          # instrumentalness: mostly near 0
    centres[:, 5] = centres[:, 5] ** 3 * 0.5    # speechiness: small
    centres[:, 11] = torch.randint(0, 114, (clusters,), device="cuda", generator=g).float() / 113.0
    which = torch.randint(0, clusters, (n,), device="cuda", generator=g)
    if contiguous:
        which = torch.sort(which).values
    t = centres[which]
    noise = torch.randn((n, 12), device="cuda", generator=g) * spread
    noise[:, [2, 4, 11]] = 0.0                  # discrete columns stay on their grid
    t = (t + noise).clamp_(0.0, 1.0)
    dup = torch.randint(0, n, (n // 50,), device="cuda", generator=g)
    src = torch.randint(0, n, (n // 50,), device="cuda", generator=g)
    t[dup] = t[src]
    return t.contiguous()

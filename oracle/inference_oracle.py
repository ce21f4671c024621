"""TEST INFRASTRUCTURE ONLY (never imported by the product): numpy restatement of the reference's sliding-window
inference arithmetic -- inference.py:115-157 (activation, sum / count accumulation), :166-210 (normals re-normalisation
or averaging where count > 0), :251-263 (uint16 / uint8 cast) -- with the patch positions of helpers.py:200-216.
`predict` is any callable mapping a (B, C, pz, py, px) float32 array to {name: (B, c, pz, py, px) logits}; the tests pass
the CPU oracle network.  **Parity unpinned**: neither inference.py nor helpers.py can be imported here (both import
zarr, which is not installed; an ordinary ModuleNotFoundError, no stand-in was written), the reference holds no test or
fixture for this path, and inference.py is broken at HEAD against its own ConfigManager.  Everything below
is a line-by-line restatement of the source text; the tests carry hand-derived known answers for the position rule."""
import numpy as np


def generate_positions(min_val, max_val, patch_size, step):
    positions = []
    pos = min_val
    while pos + patch_size <= max_val:
        positions.append(pos)
        pos += step
    last_start = max_val - patch_size
    if last_start > positions[-1]:
        positions.append(last_start)
    return sorted(set(positions))


def _activate(x, kind):
    kind = (kind or "none").lower()
    if kind == "sigmoid":
        return 1.0 / (1.0 + np.exp(-x))
    if kind == "softmax":
        e = np.exp(x - x.max(axis=1, keepdims=True))
        return e / e.sum(axis=1, keepdims=True)
    return x


def sliding_window(volume, predict, targets, patch, batch_size, positions):
    """-> ({name: blended float32}, {name: final uint8/uint16})"""
    C, Z, Y, X = volume.shape
    pz, py, px = patch
    sums = {n: np.zeros((t["channels"], Z, Y, X), np.float32) for n, t in targets.items()}
    cnts = {n: np.zeros((Z, Y, X), np.float32) for n in targets}
    for i in range(0, len(positions), batch_size):
        chunk = positions[i:i + batch_size]
        patches = np.stack([volume[:, z:z + pz, y:y + py, x:x + px] for z, y, x in chunk]).astype(np.float32)
        raw = predict(patches)
        for n, t in targets.items():
            pred = _activate(raw[n].astype(np.float32), t.get("activation", "none"))
            for b, (z, y, x) in enumerate(chunk):
                sums[n][:, z:z + pz, y:y + py, x:x + px] += pred[b]
                cnts[n][z:z + pz, y:y + py, x:x + px] += 1
    blended, final = {}, {}
    for n, t in targets.items():
        s, c = sums[n].copy(), cnts[n]
        mask = c > 0
        if n.lower() == "normals":
            if t["channels"] == 3:
                mag = np.sqrt(s[0] ** 2 + s[1] ** 2 + s[2] ** 2) + 1e-8
                for k in range(3):
                    s[k][mask] /= mag[mask]
            blended[n] = s
            v = (s + 1.0) / 2.0 * 65535.0
            final[n] = np.clip(v, 0, 65535).astype(np.uint16)
        else:
            s[..., mask] /= c[mask]
            blended[n] = s
            final[n] = np.clip(s * 255.0, 0, 255).astype(np.uint8)
    return blended, final

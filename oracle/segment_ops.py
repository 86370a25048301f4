"""TEST INFRASTRUCTURE -- CPU stand-in for pyradiomics_amd.batch.HipSegmentOps, so that the N > 1 path of
`batch.segment_matrices_sharded` (angle / z-slab / level split + one exchange step) runs under gloo without a GPU.
Same interface, torch CPU tensors in and out.  GLCM / GLRLM / GLSZM come from the C restatement (or the reference's
own cmatrices.c) through oracle.binding; the plane-range accumulators of GLDM / NGTDM are restated here in numpy
from the definitions in include/pyradiomics_amd.h (reference arithmetic: cmatrices.c:637-652, 737-748).
Never imported by the product."""
import numpy as np
import torch


class OracleSegmentOps:
    def __init__(self, cm):
        self.cm = cm

    def pair_angles(self, shape, distances, force2D, force2Ddimension):
        return self.cm.generate_angles(np.asarray(shape), list(distances), False, force2D, force2Ddimension)

    def neigh_angles(self, shape, distances, force2D, force2Ddimension):
        return self.cm.generate_angles(np.asarray(shape), list(distances), True, force2D, force2Ddimension)

    def pairs(self, image, mask, Ng, angles, force2D, force2Ddimension, fused_ok):
        return torch.from_numpy(self.cm.pairs_or_runs_for_angles("glcm", image.numpy(), mask.numpy(), Ng, 0, angles))

    def runs(self, image, mask, Ng, Nr, angles, force2D, force2Ddimension):
        return torch.from_numpy(self.cm.pairs_or_runs_for_angles("glrlm", image.numpy(), mask.numpy(), Ng, Nr, angles))

    def pairs_runs(self, image, mask, Ng, Nr, angles, force2D, force2Ddimension):
        return (self.pairs(image, mask, Ng, angles, force2D, force2Ddimension, True),
                self.runs(image, mask, Ng, Nr, angles, force2D, force2Ddimension))

    def neigh_accumulate(self, family, image, mask, Ng, z_lo, z_hi, alpha, distances, force2D, force2Ddimension):
        img = np.where(mask.numpy() != 0, image.numpy(), 0).astype(np.int64)
        angles = self.neigh_angles(img.shape, distances, force2D, force2Ddimension)
        Na = len(angles)
        reach = int(np.abs(angles).max())
        pad = np.pad(img, reach)                       # zeros = outside the mask / the volume
        centre = img[z_lo:z_hi]
        cnt = np.zeros(centre.shape, dtype=np.int64)   # valid neighbours (NGTDM) / dependence (GLDM)
        tot = np.zeros(centre.shape, dtype=np.int64)
        for a in angles:
            sl = tuple(slice(reach + int(a[d]) + (z_lo if d == 0 else 0),
                             reach + int(a[d]) + (z_hi if d == 0 else img.shape[d])) for d in range(3))
            nb = pad[sl]
            if family == 0:
                cnt += (nb > 0) & (np.abs(centre - nb) <= alpha)
            else:
                cnt += nb > 0
                tot += nb
        acc = np.zeros((Ng, Na + 1), dtype=np.int64)
        roi = centre > 0
        if (centre[roi] > Ng).any():
            raise IndexError("Calculation Failed.")
        g = centre[roi] - 1
        if family == 0:
            np.add.at(acc, (g, cnt[roi]), 1)
        else:
            np.add.at(acc, (g, 0), 1)
            c = cnt[roi]
            has = c > 0
            np.add.at(acc, (g[has], c[has]), np.abs(c[has] * centre[roi][has] - tot[roi][has]))
        return torch.from_numpy(acc)

    def neigh_finalize(self, family, acc):
        acc = acc.numpy()
        Ng, W = acc.shape
        Na = W - 1
        if family == 0:
            out = np.zeros((Ng, 2 * Na + 1))
            out[:, :W] = acc
        else:
            out = np.zeros((Ng, 3))
            out[:, 0] = acc[:, 0]
            for c in range(1, W):                       # same order as the device finalize
                out[:, 1] += acc[:, c].astype(np.float64) / float(c)
            out[:, 2] = np.arange(1, Ng + 1)
        return torch.from_numpy(out)

    def zones(self, image, mask, Ng, Ns, force2D, force2Ddimension):
        P = self.cm.calculate_glszm(image.numpy(), mask.numpy(), Ng, Ns, force2D, force2Ddimension)[0]
        keep = np.where(P.sum(axis=0) > 0)[0]
        return P[:, keep], (keep + 1).astype(np.intc)

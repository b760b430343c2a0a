"""GaLoreProjector: low-rank projection of a 2-D gradient onto the leading singular directions, refreshed every `update_proj_gap` steps.
API mirror of the reference's optim/galore_projector.py:17-124 (the GaLore authors' projector, Zhao et al. 2024): same constructor, same
`project(full_rank_grad, iter)` / `project_back(low_rank_grad)` / `get_orthogonal_matrix(weights, rank, type)`, same attributes
(`ortho_matrix` is what a checkpointed optimiser state carries).  Host logic only -- an SVD every few hundred steps and two skinny matmuls;
the arithmetic is torch's on whatever device the gradient lives on, in the order the reference applies it (pinned to the reference's outputs on
CPU tensors: tests/golden/optim_diodemix.npz)."""
import torch

_SIDES = ("left", "right", "full")


class GaLoreProjector:
    def __init__(self, rank, verbose=False, update_proj_gap=200, scale=1.0, proj_type='std'):
        self.rank = rank
        self.verbose = verbose
        self.update_proj_gap = update_proj_gap
        self.scale = scale
        self.ortho_matrix = None
        self.proj_type = proj_type

    def _side(self, shape):
        """Which factor is kept for a gradient of this shape (reference :24-52): 'std' keeps the factor of the SHORTER side (right singular
        vectors for a tall matrix), 'reverse_std' the other one; 'left' / 'right' / 'full' are fixed.  Unknown types project nothing, as there."""
        tall = shape[0] >= shape[1]
        return {"std": "right" if tall else "left", "reverse_std": "left" if tall else "right",
                "right": "right", "left": "left", "full": "full"}.get(self.proj_type)

    def project(self, full_rank_grad, iter):
        side = self._side(full_rank_grad.shape)
        if side is None:
            raise UnboundLocalError(f"GaLoreProjector: unknown proj_type {self.proj_type!r}")  # the reference falls through to an unbound local
        if self.ortho_matrix is None or iter % self.update_proj_gap == 0:
            self.ortho_matrix = self.get_orthogonal_matrix(full_rank_grad, self.rank, type=side)
        if side == "right":
            return torch.matmul(full_rank_grad, self.ortho_matrix.t())
        if side == "left":
            return torch.matmul(self.ortho_matrix.t(), full_rank_grad)
        return torch.matmul(self.ortho_matrix[0].t(), full_rank_grad) @ self.ortho_matrix[1].t()

    def project_back(self, low_rank_grad):
        # the side is re-derived from the LOW-rank shape, with the reference's own comparisons (:58-66: '>=' for std, '<=' for reverse_std)
        r, c = low_rank_grad.shape[0], low_rank_grad.shape[1]
        if self.proj_type == "std":
            side = "right" if r >= c else "left"
        elif self.proj_type == "reverse_std":
            side = "left" if r <= c else "right"
        else:
            side = self.proj_type
        if side == "right":
            full = torch.matmul(low_rank_grad, self.ortho_matrix)
        elif side == "left":
            full = torch.matmul(self.ortho_matrix, low_rank_grad)
        elif side == "full":
            full = torch.matmul(self.ortho_matrix[0], low_rank_grad) @ self.ortho_matrix[1]
        else:
            raise UnboundLocalError(f"GaLoreProjector: unknown proj_type {self.proj_type!r}")
        return full * self.scale

    def get_orthogonal_matrix(self, weights, rank, type):
        """Leading `rank` singular directions of `weights` (thin SVD in fp32; a half-precision input gets its factor back in its own dtype)."""
        if type not in _SIDES:
            raise ValueError('type should be left, right or full')
        data = weights.data
        is_float = data.dtype == torch.float
        U, _s, Vh = torch.linalg.svd(data if is_float else data.float(), full_matrices=False)
        back = (lambda t: t) if is_float else (lambda t: t.to(data.device).type(data.dtype))
        if type == "right":
            return back(Vh[:rank, :])
        if type == "left":
            return back(U[:, :rank])
        return [back(U[:, :rank]), back(Vh[:rank, :])]

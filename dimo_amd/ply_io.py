"""PLY / pth checkpoint I/O of the canonical Gaussians: the reference's on-disk interchange
format -- `save_ply / load_ply / save_model / load_model`, renderer/latent_gs_renderer.py:517-650.

The reference goes through `plyfile`; this is a dependency-free writer / reader of the same files: one `vertex`
element of `float` properties, binary little endian (what `PlyData([el]).write(path)` produces), attribute order
`x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_*` (`construct_list_of_attributes`, :517-529) and
`c_x c_y c_z c_radius` for the control points (:531-535).  ASCII PLY files are read as well.
"""
import os

import numpy as np
import torch
from torch import nn


def write_ply(path, names, table):
    """table: float array [n, len(names)]."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    table = np.ascontiguousarray(table, dtype="<f4")
    assert table.ndim == 2 and table.shape[1] == len(names)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {table.shape[0]}"]
    header += [f"property float {n}" for n in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(table.tobytes())


_PLY_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4",
              "int32": "i4", "uint": "u4", "uint32": "u4"}


def read_ply(path):
    """Returns {property name: float64 array [n]} of the first element of a PLY file."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_first, seen = None, None, [], False, 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                seen += 1
                in_first = seen == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError("list properties are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
            return {n: rows[:, i].astype(np.float64) for i, (n, _) in enumerate(props)}
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
        return {n: data[n].astype(np.float64) for n, _ in props}


class PlyMixin:
    """Mixed into `GaussianModel`."""

    def construct_list_of_attributes(self):
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += [f"f_dc_{i}" for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += [f"f_rest_{i}" for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names.append("opacity")
        names += [f"scale_{i}" for i in range(self._scaling.shape[1])]
        names += [f"rot_{i}" for i in range(self._rotation.shape[1])]
        return names

    def construct_list_of_attributes_c(self):
        return ["c_x", "c_y", "c_z", "c_radius"]

    @torch.no_grad()
    def save_ply(self, path1, path2=None):
        c = lambda t: t.detach().cpu().numpy()
        xyz = c(self._xyz)
        f_dc = c(self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
        f_rest = c(self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
        scale = c(self._r.expand_as(self._xyz)) if len(self._r) > 0 else c(self._scaling)
        table = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, c(self._opacity), scale, c(self._rotation)), axis=1)
        write_ply(path1, self.construct_list_of_attributes(), table)
        if path2 is not None:
            write_ply(path2, self.construct_list_of_attributes_c(), np.concatenate((c(self._c_xyz), c(self._c_radius)), 1))

    def load_ply(self, path1, path2=None):
        v = read_ply(path1)
        n = v["x"].shape[0]
        col = lambda prefix: [k for k in v if k.startswith(prefix)]
        xyz = np.stack((v["x"], v["y"], v["z"]), axis=1)
        features_dc = np.stack((v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]), axis=1)[:, :, None]
        extra = col("f_rest_")
        assert len(extra) == 3 * (self.max_sh_degree + 1) ** 2 - 3
        features_extra = (np.stack([v[k] for k in extra], axis=1) if extra else np.zeros((n, 0)))
        features_extra = features_extra.reshape((n, 3, (self.max_sh_degree + 1) ** 2 - 1))
        scales = np.stack([v[k] for k in col("scale_")], axis=1)
        rots = np.stack([v[k] for k in col("rot")], axis=1)
        P = lambda a: nn.Parameter(torch.tensor(a, dtype=torch.float, device=self.device).requires_grad_(True))
        self._xyz = P(xyz)
        self._features_dc = nn.Parameter(torch.tensor(features_dc, dtype=torch.float, device=self.device)
                                         .transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(torch.tensor(features_extra, dtype=torch.float, device=self.device)
                                           .transpose(1, 2).contiguous().requires_grad_(True))
        self._opacity = P(v["opacity"][:, None])
        self._scaling, self._rotation = P(scales), P(rots)
        self.active_sh_degree = self.max_sh_degree
        self.max_radii2D = torch.zeros(n, device=self.device)
        if path2 is not None:
            cv = read_ply(path2)
            self._c_xyz = P(np.stack((cv["c_x"], cv["c_y"], cv["c_z"]), axis=1))
            self._c_radius = P(cv["c_radius"][:, None])

    def save_model(self, path, step=None):
        """latent_gs_renderer.py:629-635 (latents + TimeNet state dict; the optimizer state is not saved)."""
        os.makedirs(path, exist_ok=True)
        suffix = "" if not step else f"_{step}"
        if self.vae_latent:
            torch.save({"mu": self._mu.detach().cpu(), "log_var": self._log_var.detach().cpu()},
                       os.path.join(path, f"latent_codes{suffix}.pth"))
        else:
            torch.save(self._latent_codes.detach().cpu(), os.path.join(path, f"latent_codes{suffix}.pth"))
        torch.save({k: t.detach().cpu() for k, t in self._timenet.state_dict().items()},
                   os.path.join(path, f"timenet{suffix}.pth"))

    def load_model(self, path, step=None):
        """latent_gs_renderer.py:637-650."""
        suffix = "" if not step else f"_{step}"
        lat = torch.load(os.path.join(path, f"latent_codes{suffix}.pth"), map_location=self.device)
        with torch.no_grad():
            if self.vae_latent:
                self._mu = nn.Parameter(lat["mu"].to(self.device).requires_grad_(True))
                self._log_var = nn.Parameter(lat["log_var"].to(self.device).requires_grad_(True))
            else:
                self._latent_codes = nn.Parameter(lat.to(self.device).requires_grad_(True))
        self._timenet.load_state_dict(torch.load(os.path.join(path, f"timenet{suffix}.pth"), map_location=self.device))
        self._timenet.to(self.device)
        self.max_radii2D = torch.zeros(self._xyz.shape[0], device=self.device)

"""AMReX single-level plotfile output / input for the fields of the path.

The reference writes `hydrovs` / `hydrovsbar` / noise / checkpoints with AMReX's
`WriteSingleLevelPlotfile(name, mf, varnames, geom, time, step)` (main_run_job.cpp:35-55, :406-409;
Debug.H:380-409) and reloads them with `VisMF::Read` on `<name>/Level_0/Cell` (AMReX_FileIO.H:18-34);
every validation notebook reads these directories with `yt.load`.  AMReX is not available here, so
this module writes the same on-disk layout from scratch:

    <name>/Header                  "HyperCLaw-V1.1", variable names, geometry, grid list
    <name>/Level_0/Cell_H          VisMF header: box list, FabOnDisk offsets, per-box min/max
    <name>/Level_0/Cell_D_00000    per box: "FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))<box> <ncomp>\\n" + doubles

Arrays are (ncomp, nz, ny, nx), x fastest = the FAB component-slowest layout.  The writer and the
C++ twin (include/bflbm_plotfile.H) produce identical bytes (tests/test_plotfile.py).
Not validated against yt or AMReX in this image (neither is installed): the layout follows the
published format; `read_plotfile` is this project's own reader (VisMF::Read equivalent).
"""
import os
import re

import numpy as np

FAB_DESCRIPTOR = "((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))"   # IEEE binary64, little endian


def variable_names(num_vars=22):
    """VariableNames(numVars) of the reference (AMReX_FileIO.H:209-295), the names yt shows."""
    names = ["rho", "phi", "ufx", "ufy", "ufz", "p_bulk", "ugx", "ugy", "ugz",
             "afx", "afy", "afz", "agx", "agy", "agz", "ubx", "uby", "ubz",
             "nfbarx", "ngbarx", "ufbarx", "ugbarx"]
    return names[:num_vars]


def noise_names(prefix, ncomp=19):
    """WriteOutNoise default component names fa%d / ga%d (Debug.H:394-405)."""
    return [f"{prefix}a{n}" for n in range(ncomp)]


def concatenate(root, step, ndigits=7):
    """amrex::Concatenate(root, step, Ndigits): zero-padded step suffix (main_run_job.cpp:42)."""
    return f"{root}{step:0{ndigits}d}"


def chop_boxes(n, max_grid_size):
    """BoxArray(domain).maxSize(max_grid_size): boxes in the order z-slowest, x-fastest.  When a size is not a
    multiple of max_grid_size the last box of a direction is the remainder here, whereas BoxArray::maxSize cuts
    near-equal pieces; no fixture of the reference covers that case (parity unpinned), its drivers use nx/2."""
    nx, ny, nz = n
    m = max_grid_size if max_grid_size else max(n)
    boxes = []
    for z in range(0, nz, m):
        for y in range(0, ny, m):
            for x in range(0, nx, m):
                boxes.append(((x, y, z), (min(x + m, nx) - 1, min(y + m, ny) - 1, min(z + m, nz) - 1)))
    return boxes


def _box_str(lo, hi):
    return f"(({lo[0]},{lo[1]},{lo[2]}) ({hi[0]},{hi[1]},{hi[2]}) (0,0,0))"


def _real(v):
    return format(float(v), ".17g")       # iostream precision(17), general format: "0", "1", "0.03125"


def write_plotfile(name, data, names, time=0.0, step=0, max_grid_size=None,
                   prob_lo=(0.0, 0.0, 0.0), prob_hi=(1.0, 1.0, 1.0)):
    """Write `data` (ncomp, nz, ny, nx) as a single-level plotfile directory `name`.

    `names` may be shorter than ncomp (the reference writes its 19-component checkpoints with one
    name, main_run_job.cpp:406-409; the loader ignores the Header).
    """
    data = np.ascontiguousarray(data, dtype="<f8")
    ncomp, nz, ny, nx = data.shape
    boxes = chop_boxes((nx, ny, nz), max_grid_size)
    os.makedirs(os.path.join(name, "Level_0"), exist_ok=True)
    dx = [(prob_hi[d] - prob_lo[d]) / n for d, n in enumerate((nx, ny, nz))]
    h = ["HyperCLaw-V1.1", str(len(names))] + list(names) + ["3", _real(time), "0",
         " ".join(_real(v) for v in prob_lo) + " ", " ".join(_real(v) for v in prob_hi) + " ", "",
         _box_str((0, 0, 0), (nx - 1, ny - 1, nz - 1)) + " ", f"{int(step)} ", " ".join(_real(v) for v in dx) + " ",
         "0", "0", f"0 {len(boxes)} {_real(time)}", str(int(step))]
    for lo, hi in boxes:
        for d in range(3):
            h.append(f"{_real(prob_lo[d] + lo[d] * dx[d])} {_real(prob_lo[d] + (hi[d] + 1) * dx[d])}")
    h.append("Level_0/Cell")
    with open(os.path.join(name, "Header"), "w") as fh:
        fh.write("\n".join(h) + "\n")

    offsets, mins, maxs = [], [], []
    with open(os.path.join(name, "Level_0", "Cell_D_00000"), "wb") as fd:
        for lo, hi in boxes:
            offsets.append(fd.tell())
            sub = np.ascontiguousarray(data[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1])
            fd.write(f"FAB {FAB_DESCRIPTOR}{_box_str(lo, hi)} {ncomp}\n".encode())
            fd.write(sub.tobytes())
            flat = sub.reshape(ncomp, -1)
            mins.append(flat.min(axis=1))
            maxs.append(flat.max(axis=1))
    # VisMF header: version 1, how = 1 (VisMF::NFiles, AMReX's default for a shared Cell_D_ file), ncomp, ngrow
    c = ["1", "1", str(ncomp), "0", f"({len(boxes)} 0"] + [_box_str(lo, hi) for lo, hi in boxes] + [")", str(len(boxes))]
    c += [f"FabOnDisk: Cell_D_00000 {o}" for o in offsets]
    # + 0.0: a minimum of -0.0 prints as 0 (min/max of mixed signed zeros is implementation-defined)
    c += ["", f"{len(boxes)},{ncomp}"] + [",".join(_real(v + 0.0) for v in m) + "," for m in mins]
    c += ["", f"{len(boxes)},{ncomp}"] + [",".join(_real(v + 0.0) for v in m) + "," for m in maxs]
    with open(os.path.join(name, "Level_0", "Cell_H"), "w") as fh:
        fh.write("\n".join(c) + "\n")
    return name


_BOX_RE = re.compile(r"\(\((-?\d+),(-?\d+),(-?\d+)\) \((-?\d+),(-?\d+),(-?\d+)\) \((\d+),(\d+),(\d+)\)\)")


def read_header(name):
    with open(os.path.join(name, "Header")) as fh:
        lines = fh.read().split("\n")
    nvar = int(lines[1])
    names = lines[2:2 + nvar]
    k = 2 + nvar
    dim = int(lines[k]); time = float(lines[k + 1])
    dom = _BOX_RE.search(lines[k + 6])
    hi = [int(dom.group(i)) for i in (4, 5, 6)]
    step = int(lines[k + 7].split()[0])
    ngrids = int(lines[k + 11].split()[1])
    return dict(version=lines[0], names=names, dim=dim, time=time, step=step,
                n=tuple(v + 1 for v in hi), ngrids=ngrids)


def read_plotfile(name, level=0):
    """VisMF::Read(<name>/Level_0/Cell) equivalent: returns (array (ncomp, nz, ny, nx), header dict)."""
    hdr = read_header(name)
    ldir = os.path.join(name, f"Level_{level}")
    with open(os.path.join(ldir, "Cell_H")) as fh:
        lines = fh.read().split("\n")
    ncomp = int(lines[2])
    nb = int(lines[4].strip("(").split()[0])
    boxes = []
    for i in range(nb):
        m = _BOX_RE.search(lines[5 + i])
        boxes.append(([int(m.group(j)) for j in (1, 2, 3)], [int(m.group(j)) for j in (4, 5, 6)]))
    fod = [ln for ln in lines if ln.startswith("FabOnDisk:")]
    nx, ny, nz = hdr["n"]
    out = np.empty((ncomp, nz, ny, nx))
    for (lo, hi), ln in zip(boxes, fod):
        _, fname, off = ln.split()
        with open(os.path.join(ldir, fname), "rb") as fd:
            fd.seek(int(off))
            head = fd.readline().decode()
            assert head.startswith("FAB " + FAB_DESCRIPTOR), head
            assert int(head.rsplit(" ", 1)[1]) == ncomp
            shp = (ncomp, hi[2] - lo[2] + 1, hi[1] - lo[1] + 1, hi[0] - lo[0] + 1)
            sub = np.frombuffer(fd.read(int(np.prod(shp)) * 8), dtype="<f8").reshape(shp)
        out[:, lo[2]:hi[2] + 1, lo[1]:hi[1] + 1, lo[0]:hi[0] + 1] = sub
    hdr["ncomp"] = ncomp
    hdr["boxes"] = boxes
    return out, hdr

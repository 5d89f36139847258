#!/usr/bin/env python3
"""The workflow of the reference's driver (/root/reference/tests/hierarchy_driver.cc) on the HIP back-end.

    python examples/hierarchy_driver.py -f hierarchy_input.info -d 3 -m 1 [-t 1e-6]

Same command line (`:216-253`: --filename/-f, --dim/-d, --matrix_free/-m, --tolerance/-t), same parameter handling
(`:255-283`: the INFO file is read, `fast_ap` is forced to true, `eigensolver.type` to anasazi with tolerance 1e-3,
`solver.tolerance` from the command line; matrix-free runs force `smoother.type Chebyshev`, `:400-402`), same two modes:
`"is preconditioner" false` -> 20 V-cycles on rhs = 0 from a random start vector and the convergence rate
res[20] / res[19] (`:74-101`); true -> CG preconditioned by the hierarchy (`:103-116`).  Differences: the mesh is the
hyper-cube of `laplace.n_refinements` global refinements with Q1 elements only (`laplace.fe_degree` other than 1 is
refused; where the file names none the reference takes degree 4, `:269`, this script degree 1), and a matrix-based run whose input file asks for an Ifpack relaxation (Gauss-Seidel, the file's default) takes
Jacobi, the CUDA back-end's smoother, with a note.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mfmg_amd as M  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-f", "--filename", default="hierarchy_input.info")
    ap.add_argument("-d", "--dim", type=int, default=2)
    ap.add_argument("-m", "--matrix_free", type=int, default=0)
    ap.add_argument("-t", "--tolerance", type=float, default=1e-6)
    args = ap.parse_args(argv)
    if args.dim not in (2, 3):
        raise SystemExit("dim must be 2 or 3")
    params = M.info_to_params(open(args.filename).read())
    if not args.matrix_free and params.get("use_raw_ml", False):
        raise SystemExit("use_raw_ml: ML is not part of the HIP build")
    fe_degree = int(params.get("laplace", {}).get("fe_degree", 1))
    if fe_degree != 1:
        raise SystemExit("laplace.fe_degree must be 1: higher degrees are outside the scope of the HIP build")
    params["fast_ap"] = True
    params.setdefault("eigensolver", {})["type"] = "anasazi"
    params["eigensolver"]["tolerance"] = 1e-3
    params.setdefault("solver", {})["tolerance"] = args.tolerance
    print(f"input file: {args.filename}, dimension: {args.dim}, matrix-free: {args.matrix_free}, fe_degree: {fe_degree}, "
          f"solver_tolerance: {args.tolerance}")
    if args.matrix_free:
        params.setdefault("smoother", {})["type"] = "Chebyshev"
    elif str(params.get("smoother", {}).get("type", "Jacobi")).lower() not in ("jacobi", "chebyshev"):
        print(f"note: smoother.type \"{params['smoother']['type']}\" is an Ifpack relaxation; taking Jacobi")
        params["smoother"]["type"] = "Jacobi"
    # coarse solver: the reference's default is a direct solve (Amesos / cuSOLVER); large coarse spaces take the multilevel one
    n_ref = int(params.get("laplace", {}).get("n_refinements", 5))
    cells = (2 ** n_ref,) * args.dim
    material = params.get("material_property", {}).get("type", "constant")
    if "type" not in params["solver"]:
        n_coarse = 2
        for c in cells:
            n_coarse *= max(c // int(params.get("agglomeration", {}).get("nx", 2)), 1)
        params["solver"]["type"] = "lu_dense" if n_coarse <= 4096 else "amg"
    ctx = M.Context()
    prob = M.LaplaceProblem(cells, material, device="cuda")
    evaluator = "HipMatrixFreeMeshEvaluator" if args.matrix_free else "HipMeshEvaluator"
    h = M.Hierarchy(ctx, evaluator, prob, params)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(prob.n_dofs, dtype=torch.float64, device="cuda", generator=g) * (prob.constrained == 0)
    b = torch.zeros_like(x)
    r = torch.empty_like(x)
    if not params.get("is preconditioner", True):
        def resnorm():
            h.operator_apply(0, x, r)
            ctx.sadd(r, -1.0, 1.0, b)
            return ctx.l2_norm(r)
        res = [resnorm()]
        for _ in range(20):
            h.vmult(x, b)
            res.append(resnorm())
        print(f"Convergence rate: {res[20] / res[19]:.2f}")
        rate = res[20] / res[19]
    else:
        g2 = torch.Generator(device="cuda").manual_seed(2)
        b = torch.rand(prob.n_dofs, dtype=torch.float64, device="cuda", generator=g2) * (prob.constrained == 0)
        x.zero_()
        its, hist = h.solve_cg(b, x, tolerance=args.tolerance, max_iterations=prob.n_dofs)
        print(f"Converging after {its} iterations.")
        rate = its
    print(h.timer_report())
    return rate


if __name__ == "__main__":
    main()

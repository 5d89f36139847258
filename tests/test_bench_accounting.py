"""bench.py's byte accounting (no GPU): the figures roofline.achieved is priced on."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_operator_bytes_per_dof():
    # x + y + 8 index ints + 8 coefficients
    assert bench.operator_bytes_per_dof(8, False, True) == 112
    assert bench.operator_bytes_per_dof(8, True, True) == 56
    assert bench.operator_bytes_per_dof(4, False, True) == 72
    # what the chunk-record layout requires: one id
    assert bench.operator_bytes_per_dof(8, False, False) == 84
    assert bench.operator_bytes_per_dof(8, True, False) == 28


def test_smoother_bytes_per_dof():
    # SURVEY.md 8(d), fused form: B_op + b + D^-1 (+ x_prev from the second term on)
    assert bench.smoother_bytes_per_dof(3, 8, False, True) == 128 + 136 + 136
    assert bench.smoother_bytes_per_dof(3, 8, True, True) == 72 + 80 + 80
    # required: eight coefficients per cell keep D^-1 in the records
    assert bench.smoother_bytes_per_dof(3, 8, False, False) == 100 + 108 + 108
    # one coefficient per cell: D^-1 derived in the kernel by default, stored on request
    assert bench.smoother_bytes_per_dof(3, 8, True, False) == 36 + 44 + 44
    assert bench.smoother_bytes_per_dof(3, 8, True, False, dinv_stored=True) == 44 + 52 + 52
    assert bench.smoother_bytes_per_dof(1, 8, True, False) == 36      # Jacobi


def test_survey_figures_are_the_surveys():
    # 112 B/DoF per operator application, + 32 per fused smoother term, 432 per Chebyshev(3) apply -- printed
    # under `survey_8d_*` whatever the layout of the operator at hand reads
    assert bench.survey_8d_bytes_per_dof(3) == {"operator": 112, "smoother_term": 144, "smoother_apply": 432}
    assert bench.survey_8d_bytes_per_dof(1)["smoother_apply"] == 144
    assert bench.survey_8d_bytes_per_dof(3, word=4)["operator"] == 72


def test_chebyshev_coefficients_first_term():
    c = bench.smoother_coefficients(3, 0.09, 1.8)
    assert len(c) == 3 and c[0][0] == 0.0 and abs(c[0][1] - 1.0 / 0.945) < 1e-15
    assert len(bench.smoother_coefficients(1, 0.09, 1.8)) == 1


def test_traffic_file_is_keyed_on_workload_and_tile():
    assert bench.committed_traffic(256, 3, True, (4, 3, 8)) is not None
    assert bench.committed_traffic(256, 3, True, (4, 3, 7)) is None      # another tile: no figure
    assert bench.committed_traffic(128, 3, True, (4, 3, 8)) is None      # another workload
    assert bench.committed_traffic(512, 3, False, (8, 2, 16), prefix="dofs") is not None


def test_committed_traffic_is_keyed_on_the_tiles_the_bench_ran_with():
    """roofline.traffic comes from the PMC passes committed under profiles/ (counters need rocprofv3 around the process), keyed on
    workload AND tile: the committed bench line of the round must find its entries -- a tile the library chooses differently after a
    change would silently turn the field into null."""
    import json
    line = json.loads(open(os.path.join(ROOT, "profiles", "r04_j_bench_line.json")).read())
    roof = line["roofline"]
    assert "mf_cheb_fused_kernel" in roof["kernel"]
    tile = tuple(roof["tile_waves_ty_tz"])
    traffic = bench.committed_traffic(256, 3, True, tile, prefix="sweep_cells")
    assert traffic is not None and 0.25 * roof["required_bytes_per_launch"] < traffic < roof["required_bytes_per_launch"]
    assert traffic > roof["fused_form_bytes_per_launch"]            # (the fused form is the floor of what the sweep must move)
    leg = line["north_star_512cubed_smoother"]
    assert min(leg["frac_by_layout"].values()) >= 0.5               # north_star: >= 0.50 of 8 TB/s on every layout
    gen = line["smoother_apply_512cubed_general_coefficient"]
    assert bench.committed_traffic(512, 3, False, tuple(gen["tile_waves_ty_tz"]), prefix="dofs") is not None

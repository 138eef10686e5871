"""CPU: the @options front end (ksfd_amd/options.py) -- same syntax as the reference's Parser /
SolutionParameters (KSFD/ksfdargparse.py:60-128, KSFD/ksfdsoln.py:254-347)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from ksfd_amd import options as ko_opts
from ksfd_amd.layout import cijk_to_soa

OPT2D = '@' + os.path.join(GOLDEN, 'options', 'ks2d_two_ligands.txt')
OPT1D = '@' + os.path.join(GOLDEN, 'options', 'ks1d_manufactured.txt')


def test_parse_file_petsc_block_and_dependencies():
    ns = ko_opts.parse_commandline([OPT2D, 'maxsteps=7', '--seed=5'] if False else [OPT2D, '--seed=5'])
    assert ns.save == 'solutions/ks2d' and ns.seed == 5 and ns.cappotential == 'tophat'
    assert ns.petsc[:4] == ['-ts_type', 'rosw', '-ts_adapt_type', 'basic']
    ps = ko_opts.Params(ns)
    v = ps.values0
    assert abs(v['s2'] - 0.02357 ** 2 / 2) < 1e-18          # s2=sigma**2/2 resolved through the dependency graph
    assert v['rho0'] == 9000.0 and ps.shape == (48, 48) and ps.box == (0.5, 0.5) and ps.dim == 2
    cfg = ps.problem_config()
    assert cfg.F == 3 and list(cfg.lig_group) == [0, 1]
    assert np.allclose(cfg.grp_beta, [5.56e-4, -5.56e-4]) and np.allclose(cfg.lig_D, [1e-6, 1e-5])
    o = ko_opts.step_opts_from(ps, ns.petsc)
    assert (o.adapt, o.clip_lo, o.clip_hi, o.dt_max, o.dt_min) == (1, 0.1, 5.0, 10000.0, 1e-20)
    assert (o.rtol, o.atol) == (1e-6, 0.01)
    assert ps.field_names() == ['rho', 'U_1_1', 'U_2_1']


def test_effective_ligand_defaults_are_one():
    """the defaults that take effect are alpha=beta=weight=s=gamma=D=1.0 (ksfdligand.py:578-600)"""
    ps = ko_opts.Params(ko_opts.parse_commandline(['dim=1', 'nelements=16']))
    cfg = ps.problem_config()
    assert (cfg.grp_alpha[0], cfg.grp_beta[0], cfg.lig_w[0], cfg.lig_s[0], cfg.lig_gamma[0], cfg.lig_D[0]) == (1,) * 6


def test_cycles_unknowns_duplicates_rejected():
    with pytest.raises(ValueError):
        ko_opts.Params(ko_opts.parse_commandline(['a=2*b', 'b=2*a']))
    with pytest.raises(ValueError):
        ko_opts.Params(ko_opts.parse_commandline(['a=2*zz']))
    with pytest.raises(ValueError):
        ko_opts.Params(ko_opts.parse_commandline(['dt=1', 'dt=2']))
    with pytest.raises(ValueError):
        ko_opts.step_opts_from(ko_opts.Params(ko_opts.parse_commandline(['dim=1'])), ['-ts_type', 'beuler'])


def test_fourier_series_expansion():
    """KSFD/ksfdligand.py:315-388: series_g_l=k -> k ligands, s/k, weight/k, gamma += D*(pi*i/depth)^2, s rescaled"""
    ps = ko_opts.Params(ko_opts.parse_commandline(['dim=1', 'nelements=16', 'series_1_1=3', 'depth_1_1=0.4',
                                                   's_1_1=0.01', 'gamma_1_1=0.01', 'D_1_1=1e-6', 'weight_1_1=1.5']))
    ligs = ps.ligands()
    assert len(ligs) == 3 and ps.field_names() == ['rho', 'U_1_1', 'U_1_2', 'U_1_3']
    g = [0.01 + 1e-6 * (np.pi * i / 0.4) ** 2 for i in range(3)]
    assert np.allclose([l['gamma'] for l in ligs], g) and np.allclose([l['weight'] for l in ligs], 0.5)
    assert abs(sum(l['s'] / l['gamma'] for l in ligs) - 1.0) < 1e-12       # steady-state total preserved
    assert np.allclose([l['s'] for l in ligs], ligs[0]['s'])


def test_time_dependent_parameter_and_noise_timing():
    ps = ko_opts.Params(ko_opts.parse_commandline(['dim=1', 'variance_rate=0.5', 'variance_interval=20', 's2=1e-4*(1+t)']))
    assert 's2' in ps.time_dependent() and 'variance_timing_function' in ps.time_dependent()
    assert abs(ps.values(3.0)['s2'] - 4e-4) < 1e-18
    assert ps.values(50.0)['variance_timing_function'] == 2.5


def test_manufactured_initial_state_and_source_fields_match_golden():
    """rho0/U0 expressions on x_i = i*L/n reproduce the golden u0; SpatialExpression evaluates a source like the
    reference's (KSFD/ksfdsym.py:1515-1697)."""
    z = load_golden('step_1d_manufactured')
    ns = ko_opts.parse_commandline([OPT1D])
    ps = ko_opts.Params(ns)
    cfg = ps.problem_config()
    coords = ko_opts.grid_coords(cfg)
    v = ps.values0
    u0 = np.stack([np.broadcast_to(ko_opts.SpatialExpression(ps, v[k])(0.0, coords), (128,))
                   for k in ('rho0', 'U0_1_1', 'U0_2_1')])
    assert np.allclose(u0, z['u0'], rtol=1e-15, atol=0)
    # a source written in terms of parameters and t: lamda*arho*exp(lamda*t)*sin(...) is d/dt of the exact rho
    src = ko_opts.decode_sources(['rho=lamda*arho*exp(lamda*t)*sin(2*pi*(0.25+k0*x))'], ps)
    got = src[0](2.0, coords)
    x = coords[0]
    want = v['lamda'] * np.exp(v['lamda'] * 2.0) * np.sin(2 * np.pi * (0.25 + 4.0 * x))
    assert np.allclose(got, want, rtol=1e-14)
    assert src[1].is_zero() and not src[0].is_zero()
    with pytest.raises(ValueError):
        ko_opts.decode_sources(['V=1'], ps)


REF_DIR = '/root/reference'
PARSED = os.path.join(GOLDEN, 'options_parsed.json')


def _parsed_cases():
    import json
    return sorted(json.load(open(PARSED))) if os.path.exists(PARSED) else []


@pytest.mark.skipif(not os.path.isdir(REF_DIR), reason='container-only: reads the reference\'s shipped options files')
@pytest.mark.parametrize('fname', _parsed_cases())
def test_shipped_options_files_parse_like_the_reference(fname):
    """every options file the reference ships, through ksfd_amd.options, against the table the reference's own
    parse_commandline + SolutionParameters produced for it (tests/golden/make_options_golden.py): all parameter values
    (expressions compared at sample points), ligand / group tables, the --petsc pass-through block"""
    import json
    import sympy as sy
    want = json.load(open(PARSED))[fname]
    cl = ko_opts.parse_commandline(['@' + os.path.join(REF_DIR, fname)])
    ps = ko_opts.Params(cl)
    assert list(cl.petsc) == want['petsc']
    assert cl.cappotential == want['cappotential'] and int(cl.seed) == want['seed'] and len(cl.source) == want['nsources']
    assert cl.save == want['save'] and cl.check == want['check']
    x, y, z, t = sy.symbols('x y z t')
    samples = [(0.1, 0.2, 0.3, 0.0), (0.37, 0.11, 0.05, 1.5), (0.9, 0.45, 0.6, 40.0)]
    mine = ps.values0
    for key, ref in want['values0'].items():
        assert key in mine, key
        got = mine[key]
        if isinstance(ref, dict):
            e = sy.sympify(got)
            vals = [float(e.subs({x: a, y: b, z: c, t: d})) for a, b, c, d in samples]
            assert np.allclose(vals, ref['samples'], rtol=1e-13, atol=0), key
        elif isinstance(ref, bool) or ref is None or isinstance(ref, str):
            assert (got if isinstance(got, (bool, str)) or got is None else str(got)) == ref or (ref == '' and got in ('', None, False)), (key, got, ref)
        else:
            assert abs(float(got) - float(ref)) <= 1e-14 * abs(float(ref)), (key, got, ref)
    cfg = ps.problem_config()
    ligs, groups = want['ligands'], want['groups']
    assert cfg.nlig == len(ligs) and cfg.ngroups == len(groups)
    for l, (name, g, s, gamma, D, w) in enumerate(ligs):
        assert int(cfg.lig_group[l]) == g - 1
        assert np.allclose([cfg.lig_s[l], cfg.lig_gamma[l], cfg.lig_D[l], cfg.lig_w[l]], [s, gamma, D, w], rtol=1e-14, atol=0)
    assert np.allclose(np.stack([cfg.grp_alpha, cfg.grp_beta], axis=1), groups, rtol=1e-14, atol=0)

"""The C++ adapters must compile against the reference's own headers (only checkable where
/root/reference exists; the GPU box does not have it)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference headers not present")
def test_adapters_compile_against_reference_headers():
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Wno-pragma-once-outside-header",
           "-I" + os.path.join(ROOT, "include"), "-I/root/reference/src", "-I/root/reference/includes",
           "-x", "c++", os.path.join(ROOT, "fba_pomdp_amd", "csrc", "host", "adapters.hpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_conf_bridge_compiles_and_maps_the_reference_configuration(tmp_path):
    """fba_pomdp_amd/csrc/host/conf_bridge.hpp: `to_fba_config*` instantiated with a struct that has the
    members of configurations::Conf / BAConf / FBAConf (Conf.hpp:18-34, DomainConf.hpp:14-19, PlannerConf.hpp:16-18,
    BeliefConf.hpp:16-21, BAConf.hpp:17-22, FBAConf.hpp:19 -- those headers need Boost, so the test restates the members), linked
    against the C-ABI stub: every field lands where fba_create reads it, the seed is the CLI's FNV-1a."""
    src = tmp_path / "bridge.cpp"
    src.write_text(r'''
#include <cstdio>
#include <string>
#include "conf_bridge.hpp"
struct DomainConf { std::string domain = ""; size_t size = 0, height = 0, width = 0; };
struct PlannerConf { int mcts_simulation_amount = 1000; int mcts_max_depth = -1; double mcts_exploration_const = 100; };
struct BeliefConf { size_t particle_amount = 100, resample_amount = 0; double threshold = 0; std::string option = ""; };
struct Conf { std::string seed = ""; unsigned short verbose = 0; int num_runs = 1, horizon = 10; double discount = .95;
              std::string planner = "po-uct", belief = "rejection_sampling"; PlannerConf planner_conf; DomainConf domain_conf; BeliefConf belief_conf; };
enum SAMPLETYPE { Regular, Expected };
struct BAConf : Conf { int num_episodes = 1; float noise = 0, counts_total = 10000; SAMPLETYPE bayes_sample_method = Expected; };
struct FBAConf : BAConf { std::string structure_prior = ""; };
int main() {
    FBAConf c;
    c.seed = "abc"; c.verbose = 3; c.num_runs = 7; c.horizon = 12; c.discount = .9; c.planner = "hip-po-uct"; c.belief = "hip-mh-within-gibbs";
    c.planner_conf.mcts_simulation_amount = 321; c.planner_conf.mcts_exploration_const = 50; c.domain_conf.domain = "random-collision-avoidance";
    c.domain_conf.size = 2; c.domain_conf.width = 5; c.domain_conf.height = 3; c.belief_conf.particle_amount = 77; c.belief_conf.threshold = -1.5;
    c.belief_conf.option = "rs"; c.num_episodes = 4; c.noise = .1f; c.counts_total = 40; c.bayes_sample_method = Regular; c.structure_prior = "match-uniform";
    fba_config f = fba::to_fba_config_fba(c);
    std::printf("%d %d %d %d %d %d %d %d %d %d %d %g %g %d %d %g %g %d %llu %d %d %g %d\n", f.domain, f.size, f.width, f.height, f.model, f.belief, f.planner,
                f.particles, f.sims, f.max_depth, f.horizon, f.exploration, f.discount, f.runs, f.episodes, (double)f.noise, (double)f.counts_total,
                f.structure_prior, (unsigned long long)f.seed, f.trace, f.dirichlet_regular, f.threshold, f.belief_option);
    Conf p; p.domain_conf.domain = "episodic-tiger";
    fba_config g = fba::to_fba_config(p);
    std::printf("%d %d %d %d\n", g.domain, g.model, g.belief, g.episodes);
    try { p.domain_conf.domain = "nope"; fba::to_fba_config(p); } catch (std::string const& e) { std::printf("%s\n", e.c_str()); }
    return 0;
}
''')
    exe = tmp_path / "bridge"
    cmd = ["g++", "-std=c++11", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "fba_pomdp_amd", "csrc", "host"), str(src),
           os.path.join(ROOT, "tests", "adapters", "stub_fba.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines()
    h = 1469598103934665603
    for ch in b"abc":
        h = ((h ^ ch) * 1099511628211) % 2 ** 64
    # domain 5 (random collision avoidance), size 2, 5 x 3, factored model, mh-within-gibbs (5) with option rs, po-uct ...
    assert out[0].split() == ["5", "2", "5", "3", "2", "5", "0", "77", "321", "-1", "12", "50", "0.9", "7", "4", "0.1", "40", "2", str(h), "2", "1", "-1.5", "1"]
    assert out[1].split() == ["0", "0", "0", "1"]
    assert out[2] == "please enter a legit domain, provided: nope"


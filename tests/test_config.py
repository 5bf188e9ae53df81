"""workspace.json / hparams.json (SURVEY 8 f3's remainder): the host's Config / HyperParams (csrc/host/json_gguf.cpp) against the
schema of src/core/config.cpp:30-67,121-152 — key names (config.hpp:24-26: hparams_config, model_main, model_draft), every
hparams key optional with the reference's defaults (config.hpp:33-53), n_threads clamped to the host's cores, paths joined to
the work folder.  (The reference's config.cpp itself cannot be built into oracle/_ref in this image: it needs
nlohmann::json::contains, the image has nlohmann 3.1.1 — see oracle/Makefile.  Parity of this parser is therefore pinned
to the reference's source text, not to its binary.)"""
import json
import os

import pytest


def write(d, workspace, hparams=None):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "workspace.json"), "w") as f:
        json.dump(workspace, f)
    if hparams is not None:
        with open(os.path.join(d, workspace.get("hparams_config", "hparams.json")), "w") as f:
            json.dump(hparams, f)


def test_defaults_when_nothing_is_named(tmp_path):
    from powerserve_amd import host
    d = str(tmp_path / "w")
    write(d, {})
    c = host.config_summary(d)
    assert c["batch_size"] == "128" and c["n_threads"] == "4" and c["model_main"] == "" and c["model_draft"] == ""
    assert (c["seed"], c["top_k"], c["min_keep"], c["penalty_last_n"]) == (str(2**64 - 1), "40", "0", "64")
    assert (float(c["temperature"]), float(c["top_p"]), float(c["penalty_repeat"])) == (pytest.approx(0.8), pytest.approx(0.95), 1.0)
    assert (c["penalize_nl"], c["ignore_eos"]) == ("0", "0")


def test_full_workspace(tmp_path):
    from powerserve_amd import host
    d = str(tmp_path / "w")
    hp = {"n_threads": 100000, "batch_size": 64,
          "sampler": {"seed": 1234, "temperature": 0.25, "top_p": 0.5, "top_k": 7, "min_keep": 2, "penalty_last_n": 16, "penalty_repeat": 1.5,
                      "penalty_freq": 0.125, "penalty_present": 0.0625, "penalize_nl": True, "ignore_eos": True}}
    write(d, {"hparams_config": "hp.json", "model_main": "llama-8b", "model_draft": "llama-1b", "unknown_key": 1}, hp)
    c = host.config_summary(d)
    assert c["batch_size"] == "64" and int(c["n_threads"]) == min(100000, os.cpu_count())  # clamped to hardware_concurrency
    assert c["model_main"] == os.path.join(d, "llama-8b") and c["model_draft"] == os.path.join(d, "llama-1b")
    assert (c["seed"], c["top_k"], c["min_keep"], c["penalty_last_n"], c["penalize_nl"], c["ignore_eos"]) == ("1234", "7", "2", "16", "1", "1")
    assert [float(c[k]) for k in ("temperature", "top_p", "penalty_repeat", "penalty_freq", "penalty_present")] == [0.25, 0.5, 1.5, 0.125, 0.0625]


def test_partial_sampler_section_keeps_the_other_defaults(tmp_path):
    from powerserve_amd import host
    d = str(tmp_path / "w")
    write(d, {"hparams_config": "hparams.json", "model_main": "m"}, {"sampler": {"top_k": 1}})
    c = host.config_summary(d)
    assert c["top_k"] == "1" and float(c["temperature"]) == pytest.approx(0.8) and c["batch_size"] == "128"


@pytest.mark.parametrize("seed", [2**64 - 1, 2**64 - 2, 2**53 + 1, -1, 0, 7])
def test_seed_keeps_all_64_bits(tmp_path, seed):
    """hparams "seed" is a uint64 in the reference (nlohmann reads it exactly; 2^64-1 = "pick one"): not through a double."""
    from powerserve_amd import host
    d = str(tmp_path / "w")
    write(d, {"hparams_config": "hparams.json", "model_main": "m"}, {"sampler": {"seed": seed}})
    assert host.config_summary(d)["seed"] == str(seed % 2**64)


def test_errors_are_reported(tmp_path):
    from powerserve_amd import host
    with pytest.raises(host.HostError):
        host.config_summary(str(tmp_path / "missing"))
    d = str(tmp_path / "w")
    write(d, {"hparams_config": "nope.json"})
    with pytest.raises(host.HostError):
        host.config_summary(d)

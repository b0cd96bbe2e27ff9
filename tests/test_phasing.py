"""Step 4 (GMM phasing + result files) against fixtures produced by the reference's
own split_alleles / drivers with both random sources seeded (tests/golden/make_golden.py phasing)."""
import json
import os

import numpy as np
import pytest

from nanorepeat_amd import phasing, joint
from nanorepeat_amd.round3 import Read, RepeatRegion, output_repeat_size_1d

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(HERE, "golden", "ref_phasing.json")) as f:
        return json.load(f)


def _files(tmp, stem, replace=None):
    out = {}
    for fn in sorted(os.listdir(tmp)):
        if fn.startswith(stem + ".") and fn != stem + ".fastq":
            text = open(os.path.join(tmp, fn)).read()
            if replace:
                text = text.replace(*replace)
            out[fn.replace(stem, "PREFIX")] = text
    return out


def test_phasing_1d_matches_reference(fx, tmp_path):
    for ci, case in enumerate(fx["cases_1d"]):
        stem = f"ph1_{ci}"
        rr = RepeatRegion("chr4\t3074876\t3074933\tCAG")
        rr.out_prefix = str(tmp_path / stem)
        rr.region_fq_file = str(tmp_path / (stem + ".fastq"))
        with open(rr.region_fq_file, "w") as f:
            for name, size in case["reads"]:
                rd = Read(name)
                rd.round3_repeat_size = size
                rr.read_dict[name] = rd
                f.write(f"@{name} len=8\nACGTACGT\n+\nIIIIIIII\n")
        par = case["params"]
        output_repeat_size_1d(rr)
        alleles = phasing.split_allele_using_gmm_1d(rr, par["ploidy"], par["error_rate"], par["max_mutual_overlap"],
                                                    par["max_num_components"], par["remove_noisy_reads"],
                                                    seed=case["seed"])
        assert (0 if alleles is None else len(alleles)) == case["num_alleles"], case["label"]
        assert _files(tmp_path, stem) == case["files"], case["label"]
        assert phasing.final_output_row(rr) == case["final_output"], case["label"]


def test_phasing_2d_matches_reference(fx, tmp_path):
    r1 = joint.Repeat().init_from_string("chr4:3074876:3074933:CAG:200")
    r2 = joint.Repeat().init_from_string("chr4:3074946:3074966:CCG:20")
    for ci, case in enumerate(fx["cases_2d"]):
        stem = f"ph2_{ci}"
        fq = str(tmp_path / (stem + ".fastq"))
        with open(fq, "w") as f:
            for name, a, b in case["reads"]:
                f.write(f"@{name}\nACGTACGTAC\n+\nIIIIIIIIII\n")
        counts = {name: (a, b) for name, a, b in case["reads"]}
        par = case["params"]
        phasing.split_alleles_using_gmm_2d(par["ploidy"], par["error_rate"], par["max_mutual_overlap"],
                                           par["remove_noisy_reads"], par["max_num_components"], r1, r2, counts, 0,
                                           fq, str(tmp_path / stem), seed=case["seed"])
        assert _files(tmp_path, stem, (fq, "IN.fastq")) == case["files"], case["label"]


def test_global_generators_are_the_default():
    """seed=None draws from `random` and numpy's global RandomState like the reference does."""
    import random
    counts = {f"r{i}": float(v) for i, v in enumerate([20] * 15 + [21] * 10 + [60] * 12 + [61.5] * 9)}
    random.seed(5); np.random.seed(5)
    a = phasing.phase(counts, 1, 2, 0.07, 0.15, 22, False)
    b = phasing.phase(counts, 1, 2, 0.07, 0.15, 22, False, seed=5)
    assert [x.readname_list for x in a[0]] == [x.readname_list for x in b[0]]
    assert np.allclose([x.gmm_mean1 for x in a[0]], [x.gmm_mean1 for x in b[0]], rtol=0, atol=1e-9)
    assert [x.repeat1_median_size for x in a[0]] == [20, 60]


def test_small_pieces():
    assert phasing.interval_has_overlap((0, 5), (5, 9)) and not phasing.interval_has_overlap((0, 5), (5.1, 9))
    lo, hi = phasing.get_outlier_cutoff_from_list([1.0, 1.0, 1.0, 40.0])
    assert lo == 0 and hi > 40
    names, flat = phasing.remove_outlier_reads({"a": (1.0, 2.0), "b": (1.5, 2.0)}, 2)
    assert names == ["a", "b"] and flat == [1.0, 2.0, 1.5, 2.0]
    assert phasing.data_type_error_rate("hifi") == 0.07          # the reference's `or 'clr'` quirk
    with pytest.raises(ValueError):
        phasing.data_type_error_rate("illumina")
    with pytest.raises(ValueError):
        phasing.phase({"a": 1.0, "b": 2.0}, 1, 0, 0.07, 0.15, 22, False)

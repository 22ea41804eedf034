"""The C-ABI library loads without a GPU and exports every symbol include/dmz_hip.h
declares.  No compute call is made here."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    header = open(os.path.join(ROOT, "include", "dmz_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(dmz_hip_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 20
    lib = pkg.load_library()
    for name in declared:
        assert hasattr(lib, name), "libdmz_hip.so does not export %s" % name
    assert sorted(pkg.EXPORTS) == declared
    # the test-only entry points live in their own header, outside the boundary
    test_header = open(os.path.join(ROOT, "include", "dmz_hip_test.h")).read()
    test_declared = sorted(set(re.findall(r"\b(dmz_hip_[a-z0-9_]+)\s*\(", test_header)))
    assert test_declared == sorted(pkg.TEST_EXPORTS) and not set(test_declared) & set(declared)
    for name in test_declared:
        assert hasattr(lib, name), "libdmz_hip.so does not export %s" % name


def test_result_record_layout(pkg, orc):
    assert pkg.RESULT_DTYPE.itemsize == 1024
    assert pkg.RESULT_DTYPE == orc.RESULT_DTYPE
    header = open(os.path.join(ROOT, "include", "dmz_hip.h")).read()
    assert "uint8_t reserved[1024 - 816];" in header
    assert pkg.RESULT_DTYPE.fields["reserved"][1] == 816


def test_expiry_and_session_record_layouts(pkg, orc):
    assert pkg.EXPIRY_DTYPE == orc.EXPIRY_DTYPE and pkg.EXPIRY_DTYPE.itemsize == 1592
    assert pkg.SESSION_DTYPE == orc.SESSION_DTYPE and pkg.SESSION_DTYPE.itemsize == 128
    header = open(os.path.join(ROOT, "include", "dmz_hip.h")).read()
    assert "#define DMZ_HIP_EXPIRY_MAX_GROUPS 8" in header and "/* 1592 bytes */" in header and "/* 128 bytes */" in header


def test_no_gpu_means_loud_failure(pkg):
    import pytest
    lib = pkg.load_library()
    if lib.dmz_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.DmzHipError):
        pkg.Context(0)


def test_product_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "card.io-dmz_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "import orc" not in text and "dmz_oracle.h" not in text, f


def test_bench_tables_cover_every_stage(pkg):
    """bench.py's algorithmic-bytes and PMC-traffic tables name exactly the stages the C-ABI times"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert set(bench.ALGO) == set(pkg.STAGES) == set(bench.STAGE_KERNELS)
    header = open(os.path.join(ROOT, "include", "dmz_hip.h")).read()
    assert "#define DMZ_HIP_STAGE_COUNT %d" % len(pkg.STAGES) in header
    # algorithmic bytes per unit of configs[1..3]
    assert [bench.CONFIGS[c]["bytes"] for c in (2, 3, 4)] == [307280, 116584, 423784]
    for c in bench.CONFIGS.values():
        assert set(c["stages"]) <= set(pkg.STAGES)


def test_capi_shard_ranges_tile_the_corpus_like_sharding_py(pkg):
    """dmz_hip_shard_range (no device needed): contiguous, in rank order, covering [0, n) exactly once, and the same split
    as card.io-dmz_amd/sharding.py uses for the torch.distributed path -- a C++ host and bench.py shard alike."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("dmz_sharding", os.path.join(os.path.dirname(pkg.__file__), "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    for n in (0, 1, 7, 4096, 65536, 1048576, 1000003, (1 << 40) + 12345):
        for world in (1, 2, 3, 4, 8, 13):
            pos = 0
            for rank in range(world):
                first, count = pkg.shard_range(n, world, rank)
                assert first == pos and count >= 0
                assert (first, first + count) == sharding.shard_range(n, rank, world)
                pos += count
            assert pos == n
    assert pkg.shard_range(100, 0, 0) == (0, 0) and pkg.shard_range(100, 4, 7) == (0, 0)  # bad requests: empty range

"""The oracle must reproduce the reference's own golden tables — this is what pins it."""
import pytest

import kai_testlib as T

FILES = ["allocate__allocate", "allocate__allocateFractionalGpu", "allocate__allocateGpuMemory", "allocate__allocateMIG", "allocate__allocateGang", "allocate__allocateElastic", "allocate__allocate_subgroups", "allocate__allocateTopology",
         "reclaim__reclaimGpuMemory", "reclaim__reclaimMIG", "preempt__preemptGpuMemory", "preempt__preemptMIG", "consolidation__consolidationGpuMemory",
         "reclaim__reclaim", "reclaim__reclaimDepartments", "reclaim__reclaimGang", "reclaim__reclaim_elastic", "reclaim__reclaim_sub_group",
         "preempt__preempt", "preempt__preemptGang", "preempt__preempt_elastic", "preempt__preempt_subgroups",
         "consolidation__consolidation", "consolidation__consolidation_subgroups"]


def _cases(name):
    doc = T.load_golden(name)
    return [(name, i, c, doc["actions"]) for i, c in enumerate(doc["cases"])]


ALL = [x for f in FILES for x in _cases(f)]


@pytest.mark.parametrize("name,i,case,actions", ALL, ids=[f"{n}[{i}]" for n, i, _, _ in ALL])
def test_oracle_reproduces_reference_expectations(name, i, case, actions):
    try:
        snap, cfg, meta = T.case_to_snapshot(case, fractions=True)
    except T.Unsupported as e:
        pytest.skip(f"outside the built path: {e}")
    res = T.Oracle.run(snap, cfg, actions)
    errs = T.check_expectations(snap, meta, res.pod_status, res.pod_node, res.nodes, res.gpu_groups)
    assert not errs, f"{meta['name']} ({name} line {meta['line']}): {errs}"


def test_golden_coverage():
    """The scenarios must be exercised, not silently skipped: 250 of the 251 action-test scenarios (99 allocate, 22 of them with fractional GPUs, 6 with
    GPU-memory requests, 6 with MIG; 151 reclaim / preempt / consolidation, 7 of them with shared GPUs, 14 with GPU-memory requests, 7 with MIG); the one
    left out is a fixture built by Go code instead of a literal."""
    ok = 0
    for name, i, case, actions in ALL:
        try:
            T.case_to_snapshot(case, fractions=True)
            ok += 1
        except T.Unsupported:
            pass
    assert ok >= 250, ok


INTEG_FILES = ("integration_tests__allocate__allocate", "integration_tests__allocate__allocate_topology", "integration_tests__reclaim__reclaim",
               "integration_tests__preempt__preempt", "integration_tests__preempt__preemptGang", "integration_tests__consolidation__consolidation",
               "integration_tests__consolidation__consolidationGang", "integration_tests__consolidation_and_reclaim__consolidation_and_reclaim",
               "integration_tests__allocate__allocateFractionalGpu", "integration_tests__allocate__allocateMIG", "integration_tests__consolidation__consolidationFractional",
               "integration_tests__preempt__preemptFractional", "integration_tests__preempt__preemptMIG", "integration_tests__reclaim__reclaimFractional",
               "integration_tests__reclaim__reclaimMIG")
INTEG = [(n, i, c) for n in INTEG_FILES for i, c in enumerate(T.load_golden(n)["cases"])]


@pytest.mark.parametrize("name,i,case", INTEG, ids=[f"{n}[{i}]" for n, i, _ in INTEG])
def test_oracle_reproduces_integration_expectations(name, i, case):
    """The reference's integration tests (actions/integration_tests/*): several scheduling cycles — allocate, consolidation, reclaim,
    preempt on a session rebuilt each round with the outcome fed back — must end in the expected cluster state and stay there."""
    try:
        errs = T.run_integration(case, T.Oracle.run, fractions=True)
    except T.Unsupported as e:
        pytest.skip(f"outside the built path: {e}")
    assert not errs, f"{case.get('Name')} ({name} line {case.get('_line')}): {errs[:4]}"


def test_integration_coverage():
    ok = 0
    for name, i, case in INTEG:
        try:
            T.run_integration(case, T.Oracle.run, rounds_after=0, fractions=True); ok += 1
        except T.Unsupported:
            pass
    assert ok >= 113, ok

"""The facade's fork-join pool (similari_amd/csrc/sa_pool.h), compiled for the host: every job exactly once on its thread, and the CPUs
two pools of one process bind their workers to never overlap (two trackers created by one thread)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = r"""
#include "sa_pool.h"
#include <cstdio>
#include <set>
#include <stdexcept>
int main() {
  int allowed = 0;
  cpu_set_t set; CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0) allowed = CPU_COUNT(&set);
  SaPool a(2), b(2);
  SaPool* c = new SaPool(1);
  std::set<int> seen;
  size_t n = 0;
  for (const SaPool* p : {(const SaPool*)&a, (const SaPool*)&b, (const SaPool*)c})
    for (int cpu : p->claimed_cpus()) { seen.insert(cpu); ++n; }
  const bool pinned = !a.claimed_cpus().empty();
  if (allowed >= 9 && !(pinned && a.claimed_cpus().size() == 3 && b.claimed_cpus().size() == 3 && c->claimed_cpus().size() == 2)) { printf("BAD sizes\n"); return 1; }
  if (seen.size() != n) { printf("BAD overlap\n"); return 1; }
  if (pinned && (a.next_cpu() != a.claimed_cpus().back())) { printf("BAD next\n"); return 1; }
  const std::vector<int> was = c->claimed_cpus();
  delete c;                               // its CPUs are free again: the next pool takes them
  SaPool d(1);
  if (!was.empty() && d.claimed_cpus() != was) { printf("BAD release\n"); return 1; }
  std::vector<std::atomic<int>> hits(64);
  for (int rep = 0; rep < 2000; ++rep) {
    const uint32_t jobs = 1 + rep % 64;
    for (auto& h : hits) h = 0;
    a.run(jobs, [&](uint32_t i) { hits[i].fetch_add(1); });
    for (uint32_t i = 0; i < 64; ++i) if (hits[i] != (i < jobs ? 1 : 0)) { printf("BAD job %u of %u\n", i, jobs); return 1; }
  }
  // a job that throws: run() comes back (every worker has answered) and rethrows on the caller; the pool works afterwards
  for (uint32_t bad : {0u, 1u, 5u}) {
    bool caught = false;
    try { a.run(8, [&](uint32_t i) { if (i == bad) throw std::runtime_error("job failed"); }); } catch (const std::runtime_error&) { caught = true; }
    if (!caught) { printf("BAD exception of job %u lost\n", bad); return 1; }
    for (auto& h : hits) h = 0;
    a.run(8, [&](uint32_t i) { hits[i].fetch_add(1); });
    for (uint32_t i = 0; i < 8; ++i) if (hits[i] != 1) { printf("BAD run after an exception\n"); return 1; }
  }
  SaPool lazy(2, true, 0);   // spin_us = 0: the workers sleep at once
  for (int rep = 0; rep < 50; ++rep) {
    for (auto& h : hits) h = 0;
    lazy.run(6, [&](uint32_t i) { hits[i].fetch_add(1); });
    for (uint32_t i = 0; i < 6; ++i) if (hits[i] != 1) { printf("BAD lazy pool\n"); return 1; }
  }
  printf("ok allowed=%d pinned=%d\n", allowed, (int)pinned);
  return 0;
}
"""


def test_pools_of_one_process_claim_disjoint_cpus_and_run_every_job_once(tmp_path):
    if sys.platform != "linux":
        pytest.skip("sched_setaffinity")
    src = tmp_path / "pool_claims.cpp"
    src.write_text(SRC)
    exe = tmp_path / "pool_claims"
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-I", str(ROOT / "similari_amd" / "csrc"), str(src), "-o", str(exe)], check=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr

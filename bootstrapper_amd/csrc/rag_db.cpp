// The RAG database of `bs segment --ws` (blockwise) written natively.
//
// The reference keeps the fragment graph in a database that its blockwise tasks fill row by row (volara's SQLite / PostgreSQL
// graph store, configured by the `db` table of the segment config: /root/reference/bootstrapper/post/watershed.py:100-117,
// post/blockwise/watershed_frags.py:230-246 nodes {position, size}, waterz_agglom.py:165-170 edges {merge_score}).  Here the
// graph of a volume exists in host memory when the block stages end, and is exported once: 937 000 nodes and 1.24 million edges
// for the 1024^3 benchmark volume.  Through Python's sqlite3 module that export was 3.1 s (a tuple per row) and the last thing
// `bs segment` waited for; through the SQLite C API (libsqlite3.so.0, loaded at run time like libzstd) it is prepared statements,
// bound and stepped in a loop, rows in primary-key order.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/bsmi_io.h"

namespace bsmi {
void set_error(const char* fmt, ...);
}

namespace {

struct Sqlite {
  void* so = nullptr;
  int (*open)(const char*, void**) = nullptr;
  int (*close)(void*) = nullptr;
  int (*exec)(void*, const char*, int (*)(void*, int, char**, char**), void*, char**) = nullptr;
  int (*prepare)(void*, const char*, int, void**, const char**) = nullptr;
  int (*bind_int64)(void*, int, long long) = nullptr;
  int (*bind_double)(void*, int, double) = nullptr;
  int (*bind_null)(void*, int) = nullptr;
  int (*step)(void*) = nullptr;
  int (*reset)(void*) = nullptr;
  int (*finalize)(void*) = nullptr;
  const char* (*errmsg)(void*) = nullptr;
  bool ok = false;
};

const Sqlite* sqlite_api() {
  static const Sqlite api = [] {
    Sqlite a;
    for (const char* name : {"libsqlite3.so.0", "libsqlite3.so"}) {
      a.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (a.so) break;
    }
    if (!a.so) return a;
    auto sym = [&](const char* n) { return dlsym(a.so, n); };
    a.open = (decltype(a.open))sym("sqlite3_open");
    a.close = (decltype(a.close))sym("sqlite3_close");
    a.exec = (decltype(a.exec))sym("sqlite3_exec");
    a.prepare = (decltype(a.prepare))sym("sqlite3_prepare_v2");
    a.bind_int64 = (decltype(a.bind_int64))sym("sqlite3_bind_int64");
    a.bind_double = (decltype(a.bind_double))sym("sqlite3_bind_double");
    a.bind_null = (decltype(a.bind_null))sym("sqlite3_bind_null");
    a.step = (decltype(a.step))sym("sqlite3_step");
    a.reset = (decltype(a.reset))sym("sqlite3_reset");
    a.finalize = (decltype(a.finalize))sym("sqlite3_finalize");
    a.errmsg = (decltype(a.errmsg))sym("sqlite3_errmsg");
    a.ok = a.open && a.close && a.exec && a.prepare && a.bind_int64 && a.bind_double && a.bind_null && a.step && a.reset && a.finalize && a.errmsg;
    return a;
  }();
  return &api;
}

constexpr int kSqliteDone = 101;

}  // namespace

extern "C" int bsmi_rag_write_sqlite(const char* path, uint64_t n_nodes, const uint64_t* ids, const double* positions, const int64_t* sizes,
                                     uint64_t n_edges, const uint64_t* edges, const float* scores) {
  if (!path || (n_nodes && (!ids || !positions || !sizes)) || (n_edges && (!edges || !scores))) {
    bsmi::set_error("bsmi_rag_write_sqlite: null argument");
    return BSMI_ERR_INVALID;
  }
  const Sqlite* q = sqlite_api();
  if (!q->ok) {
    bsmi::set_error("libsqlite3.so.0 is not available");
    return BSMI_ERR_MISSING;
  }
  void* db = nullptr;
  if (q->open(path, &db) != 0) {
    bsmi::set_error("%s: %s", path, db ? q->errmsg(db) : "cannot open");
    if (db) q->close(db);
    return BSMI_ERR_INVALID;
  }
  auto fail = [&](const char* what) {
    bsmi::set_error("%s: %s: %s", path, what, q->errmsg(db));
    q->exec(db, "ROLLBACK", nullptr, nullptr, nullptr);
    q->close(db);
    return BSMI_ERR_INVALID;
  };
  // a file written once from scratch: no rollback journal, no fsync, one transaction
  const char* setup =
      "PRAGMA page_size = 32768; PRAGMA journal_mode = OFF; PRAGMA synchronous = OFF; PRAGMA cache_size = -400000; PRAGMA locking_mode = EXCLUSIVE;"
      "BEGIN;"
      "DROP TABLE IF EXISTS nodes; DROP TABLE IF EXISTS edges;"
      "CREATE TABLE nodes (id INTEGER PRIMARY KEY, z REAL, y REAL, x REAL, size INTEGER);"
      "CREATE TABLE edges (u INTEGER, v INTEGER, merge_score REAL, PRIMARY KEY (u, v)) WITHOUT ROWID;";   // one tree, not a table + an index
  if (q->exec(db, setup, nullptr, nullptr, nullptr) != 0) return fail("schema");
  // rows in key order: sequential inserts into the primary-key trees
  std::vector<uint32_t> order(n_nodes);
  std::iota(order.begin(), order.end(), 0u);
  if (!std::is_sorted(ids, ids + n_nodes)) std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ids[a] < ids[b]; });
  // kRows rows per statement (one pass through SQLite's virtual machine per statement, not per row: 1.4 -> 0.8 s for the 2.2
  // million rows of the 1024^3 volume), the remainder row by row
  constexpr int kRows = 16;
  auto values = [](const char* head, const char* row, int rows) {
    std::string sql = head;
    for (int r = 0; r < rows; ++r) sql += (r ? "," : "") + std::string(row);
    return sql;
  };
  for (int pass = 0; pass < 2; ++pass) {
    const int rows = pass == 0 ? kRows : 1;
    const uint64_t begin = pass == 0 ? 0 : n_nodes / kRows * kRows, end = pass == 0 ? n_nodes / kRows * kRows : n_nodes;
    if (begin == end) continue;
    void* st = nullptr;
    if (q->prepare(db, values("INSERT INTO nodes VALUES ", "(?,?,?,?,?)", rows).c_str(), -1, &st, nullptr) != 0) return fail("prepare nodes");
    for (uint64_t k = begin; k < end; k += rows) {
      for (int r = 0; r < rows; ++r) {
        const uint32_t i = order[k + r];
        q->bind_int64(st, 5 * r + 1, (long long)ids[i]);
        q->bind_double(st, 5 * r + 2, positions[3 * (size_t)i]);
        q->bind_double(st, 5 * r + 3, positions[3 * (size_t)i + 1]);
        q->bind_double(st, 5 * r + 4, positions[3 * (size_t)i + 2]);
        q->bind_int64(st, 5 * r + 5, (long long)sizes[i]);
      }
      if (q->step(st) != kSqliteDone) { q->finalize(st); return fail("insert into nodes"); }
      q->reset(st);
    }
    q->finalize(st);
  }
  order.resize(n_edges);
  std::iota(order.begin(), order.end(), 0u);
  auto before = [&](uint32_t a, uint32_t b) {
    return edges[2 * (size_t)a] != edges[2 * (size_t)b] ? edges[2 * (size_t)a] < edges[2 * (size_t)b] : edges[2 * (size_t)a + 1] < edges[2 * (size_t)b + 1];
  };
  if (!std::is_sorted(order.begin(), order.end(), before)) std::sort(order.begin(), order.end(), before);
  for (uint64_t k = 1; k < n_edges; ++k)   // (one statement holds many rows: a duplicate pair would fail all of them)
    if (edges[2 * (size_t)order[k]] == edges[2 * (size_t)order[k - 1]] && edges[2 * (size_t)order[k] + 1] == edges[2 * (size_t)order[k - 1] + 1]) {
      bsmi::set_error("%s: edge (%llu, %llu) is listed twice", path, (unsigned long long)edges[2 * (size_t)order[k]], (unsigned long long)edges[2 * (size_t)order[k] + 1]);
      q->exec(db, "ROLLBACK", nullptr, nullptr, nullptr);
      q->close(db);
      return BSMI_ERR_INVALID;
    }
  for (int pass = 0; pass < 2; ++pass) {
    const int rows = pass == 0 ? kRows : 1;
    const uint64_t begin = pass == 0 ? 0 : n_edges / kRows * kRows, end = pass == 0 ? n_edges / kRows * kRows : n_edges;
    if (begin == end) continue;
    void* st = nullptr;
    if (q->prepare(db, values("INSERT INTO edges VALUES ", "(?,?,?)", rows).c_str(), -1, &st, nullptr) != 0) return fail("prepare edges");
    for (uint64_t k = begin; k < end; k += rows) {
      for (int r = 0; r < rows; ++r) {
        const uint32_t i = order[k + r];
        q->bind_int64(st, 3 * r + 1, (long long)edges[2 * (size_t)i]);
        q->bind_int64(st, 3 * r + 2, (long long)edges[2 * (size_t)i + 1]);
        if (std::isnan(scores[i])) q->bind_null(st, 3 * r + 3);
        else q->bind_double(st, 3 * r + 3, (double)scores[i]);
      }
      if (q->step(st) != kSqliteDone) { q->finalize(st); return fail("insert into edges"); }
      q->reset(st);
    }
    q->finalize(st);
  }
  if (q->exec(db, "COMMIT", nullptr, nullptr, nullptr) != 0) return fail("commit");
  q->close(db);
  return BSMI_OK;
}

"""Host-side mirror of `SearchProvider` (src/search/search_provider.rs:66-333) over the C ABI.

Same method names, argument meaning and error behaviour as the reference; the only piece that is NOT
mirrored is the SQLite page store (metadata, no vector math — out of scope): pages are kept
in an in-memory table keyed by the same monotone 1-based rowid SQLite would hand out (:275-277).
The Rust shim in INTEGRATION.md is this file written against `extern "C"`.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

from ._lib import DawnError, EM_LEN, NotNormalizedError
from .index import BestResults, VectorIndex, is_normalized

TOP_K = 20  # search_provider.rs:214
MAX_LOCAL_PAGES = 1_000_000  # search_provider.rs:164-166


@dataclass
class ExtractedPage:  # src/search/page_source.rs ExtractedPage{url,title,text,combined}
    url: str
    title: str = ""
    text: str = ""
    combined: str = ""


@dataclass
class FoundPage:  # search_provider.rs:51-59
    instance_id: str
    page_id: int
    distance: float
    url: str
    title: str
    text: str


@dataclass
class SearchResult:  # search_provider.rs:44-49
    pages: List[FoundPage] = field(default_factory=list)
    servers_contacted: int = 0
    pages_searched: int = 0


@dataclass
class SearchStats:  # search_provider.rs:61-64
    pages_indexed: int = 0


class SearchProvider:
    def __init__(self, device: int = 0):
        self.index = VectorIndex(device)  # new_index(&INDEX_OPTIONS) :102
        self._pages: Dict[int, tuple] = {}  # id -> (url, title, text, embedding f32[384])
        self._by_url: Dict[str, int] = {}
        self._next_rowid = 1

    # -- :155-166 ---------------------------------------------------------------------------------
    def page_count(self) -> int:
        return len(self._pages)

    def local_space_available(self) -> bool:
        return self.page_count() < MAX_LOCAL_PAGES

    # -- :183-200 ---------------------------------------------------------------------------------
    def embedding_for_page(self, id_: int) -> np.ndarray:
        if id_ not in self._pages:
            raise KeyError(f"Page not found in DB: {id_}")
        return self._pages[id_][3].copy()

    def search_like(self, id_: int) -> SearchResult:
        return self.search_embedding(self.embedding_for_page(id_))

    # -- :202-248 ---------------------------------------------------------------------------------
    def search_embedding(self, query_embedding: np.ndarray) -> SearchResult:
        q = np.ascontiguousarray(query_embedding, dtype=np.float32)
        if q.shape != (EM_LEN,):
            raise ValueError("query must have 384 elements")  # try_into()? failure :206
        if not is_normalized(q):
            raise NotNormalizedError(-2, "Search vector is not normalized")
        labels, distances = self.index.search(q, TOP_K)
        pages = []
        for distance, id_ in zip(distances, labels):
            row = self._pages.get(int(id_))
            if row is None:
                print(f"Page not found in DB: {id_}")  # :238
                continue
            pages.append(FoundPage("", int(id_), float(distance), row[0], row[1], row[2]))
        return SearchResult(pages=pages, servers_contacted=0, pages_searched=self.index.size())

    # -- :250-286 ---------------------------------------------------------------------------------
    def insert(self, page: ExtractedPage, q: np.ndarray) -> None:
        if not self.local_space_available():
            raise DawnError(-1, "No space available")
        if page.url in self._by_url:
            print(f"Already have with id {page.url}")  # :261
            return
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.shape != (EM_LEN,) or not is_normalized(q):
            raise NotNormalizedError(-2, "Insert embedding is not normalized")
        id_ = self._next_rowid  # last_insert_rowid() :275-277
        self._next_rowid += 1
        self._pages[id_] = (page.url, page.title, page.text, q.copy())
        self._by_url[page.url] = id_
        if self.index.size() == self.index.capacity():  # :280-283
            self.index.reserve(self.index.size() + 1024)
        self.index.add(id_, q)

    def insert_batch(self, pages: List[ExtractedPage], embeddings: np.ndarray) -> None:
        """`insert` for many pages in one call (what a batching indexer binds: one dawn_index_add_batch instead of one
        dawn_index_add per page).  Same gates, same rowids; a page whose URL is already stored is skipped (:259-263); a
        non-normalised embedding fails the whole call before anything is stored."""
        emb = np.ascontiguousarray(embeddings, dtype=np.float32)
        if emb.ndim != 2 or emb.shape != (len(pages), EM_LEN):
            raise ValueError("embeddings must be [len(pages), 384]")
        keep = [i for i, p in enumerate(pages) if p.url not in self._by_url]
        if self.page_count() + len(keep) > MAX_LOCAL_PAGES:
            raise DawnError(-1, "No space available")
        for i in keep:
            if not is_normalized(emb[i]):
                raise NotNormalizedError(-2, "Insert embedding is not normalized")
        if not keep:
            return
        ids = np.arange(self._next_rowid, self._next_rowid + len(keep), dtype=np.uint64)
        if self.index.size() + len(keep) > self.index.capacity():
            self.index.reserve(self.index.size() + max(len(keep), 1024))
        self.index.add_batch(ids, emb[keep])
        for id_, i in zip(ids.tolist(), keep):
            self._pages[id_] = (pages[i].url, pages[i].title, pages[i].text, emb[i].copy())
            self._by_url[pages[i].url] = id_
        self._next_rowid += len(keep)

    # -- :127-153 ---------------------------------------------------------------------------------
    def fill_index_from_db(self) -> None:
        count = self.page_count()
        self.index.reserve(count)
        if count:
            ids = np.fromiter(self._pages.keys(), dtype=np.uint64, count=count)
            rows = np.stack([self._pages[int(i)][3] for i in ids])
            self.index.add_batch(ids, rows)

    def save(self, path: str) -> None:  # :173-181
        self.index.save(path)

    def stats(self) -> SearchStats:  # :328-332
        return SearchStats(pages_indexed=self.page_count())


def search_remote_merge(local: SearchResult, remote_pages: List[FoundPage], remote_pages_searched: int = 0,
                        servers_contacted: int = 0):
    """The merge half of SearchService::search_remote (src/search/search_service.rs:201-277).

    Returns (distance_limit sent to peers, merged SearchResult).  `distance_limit` is
    BestResults::worst_distance() of the local results — 0.0 until 20 local results exist (:222 with
    best_results.rs:40), exactly as the reference sends it."""
    all_found = list(local.pages)
    best = BestResults(TOP_K)  # :214
    for i, page in enumerate(all_found):
        best.insert(i, page.distance)
    worst_distance = best.worst_distance()  # :222
    for x in remote_pages:  # :246-259
        best.insert(len(all_found), x.distance)
        all_found.append(x)
    best.sort()  # :262
    real = [all_found[i] for i, _ in best.results()]
    return worst_distance, SearchResult(pages=real, pages_searched=local.pages_searched + remote_pages_searched,
                                        servers_contacted=servers_contacted)

"""TEST INFRASTRUCTURE -- CPU restatement of the reference's database_context (which keyframes of a tiered
compressed_database are resident, and the runtime metadata seek_v0 reads), in numpy.

Follows /root/reference/includes/acl:
  compressed_database layout            core/impl/compressed_headers.h:447-606
  database_context::initialize          decompression/database/impl/database.impl.h:100-260
  stream_in / stream_out chunk choice   decompression/database/impl/database.impl.h:443-640
  tier metadata publish / retire        decompression/database/impl/database.impl.h:166-201, 595-620
  in-memory streamer                    decompression/database/impl/debug_database_streamer.h:60-110

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes

import numpy as np

from . import bindings

TAG_COMPRESSED_DATABASE = 0xAC11DB01
TIER_MEDIUM, TIER_LOW = 1, 2


def _u32(buffer, offset, count=1):
    return np.frombuffer(buffer, dtype="<u4", count=count, offset=offset)


class OracleDatabase:
    """Mirrors database_context<default_database_settings> bound to in-memory streamers.

    `runtime_headers` is the [clip header | segment headers...] block of core/impl/compressed_headers.h:397-439 and
    `resident[tier]` the streamer's destination buffer (poisoned where nothing was streamed in, like the debug streamer).
    """

    def __init__(self, database, bulk_medium=None, bulk_low=None):
        data = bytes(np.asarray(database, dtype=np.uint8).tobytes())
        size, _hash = (int(v) for v in _u32(data, 0, 2))
        base = 8  # database_header follows the raw_buffer_header
        tag = int(_u32(data, base)[0])
        if tag != TAG_COMPRESSED_DATABASE:
            raise ValueError("not a compressed_database")
        version, misc = (int(v) for v in np.frombuffer(data, dtype="<u2", count=2, offset=base + 4))
        (n_medium, n_low, self.max_chunk_size, self.num_clips, self.num_segments, clip_metadata_offset,
         size_medium, size_low, offset_medium, offset_low, _hash_medium, _hash_low) = (int(v) for v in _u32(data, base + 8, 12))
        self.version = version
        self.is_bulk_data_inline = (misc & 1) != 0
        self.num_chunks = {TIER_MEDIUM: n_medium, TIER_LOW: n_low}
        self.bulk_data_size = {TIER_MEDIUM: size_medium, TIER_LOW: size_low}

        sources = {TIER_MEDIUM: bulk_medium, TIER_LOW: bulk_low}
        inline_offsets = {TIER_MEDIUM: offset_medium, TIER_LOW: offset_low}
        self.source = {}
        for tier in (TIER_MEDIUM, TIER_LOW):
            if sources[tier] is not None:
                self.source[tier] = np.asarray(sources[tier], dtype=np.uint8)[: self.bulk_data_size[tier]].copy()
            elif self.bulk_data_size[tier] == 0:
                self.source[tier] = np.zeros(0, dtype=np.uint8)
            else:
                if not self.is_bulk_data_inline:
                    raise ValueError("bulk data is neither inline nor provided")
                start = base + inline_offsets[tier]
                self.source[tier] = np.frombuffer(data, dtype=np.uint8, count=self.bulk_data_size[tier], offset=start).copy()

        # chunk descriptions: {size, offset} pairs, medium tier first (database_header::get_chunk_descriptions_*)
        descriptions = _u32(data, base + 56, 2 * (n_medium + n_low)).reshape(-1, 2)
        self.chunks = {TIER_MEDIUM: descriptions[:n_medium].astype(np.int64), TIER_LOW: descriptions[n_medium:].astype(np.int64)}

        clip_metadata = _u32(data, base + clip_metadata_offset, 2 * self.num_clips).reshape(-1, 2)
        self.clip_hashes = [int(h) for h in clip_metadata[:, 0]]

        # runtime headers: zero = nothing streamed in; clip hashes are set up front (database.impl.h:151-157)
        self.runtime_headers = np.zeros(max(self.num_clips * 8 + self.num_segments * 16, 16), dtype=np.uint8)
        for clip_hash, clip_header_offset in clip_metadata:
            self.runtime_headers[int(clip_header_offset): int(clip_header_offset) + 4] = np.frombuffer(np.uint32(clip_hash).tobytes(), dtype=np.uint8)

        # +16: the decoder reads whole 8 byte windows that may start in the last bytes of a chunk's samples
        self.resident = {tier: np.full(self.bulk_data_size[tier] + 16, 0xCD, dtype=np.uint8) for tier in (TIER_MEDIUM, TIER_LOW)}
        self.loaded = {tier: [False] * self.num_chunks[tier] for tier in (TIER_MEDIUM, TIER_LOW)}
        self._binding = None

    def contains(self, clip_blob):
        """compressed_database::contains (core/impl/compressed_database.impl.h:123-140)"""
        blob = np.asarray(clip_blob, dtype=np.uint8)
        misc = int(_u32(blob.tobytes(), 8 + 20)[0])
        if (misc & (1 << 8)) == 0:
            return False
        return int(_u32(blob.tobytes(), 4)[0]) in self.clip_hashes

    def is_streamed_in(self, tier):
        return all(self.loaded[tier])

    # ---- streaming -------------------------------------------------------------------------------------------
    def _first_chunk(self, tier, stream_in):
        # database.impl.h:478-497 / 551-570: the bitset is scanned one 32 bit entry at a time, first chunk in the MSB. Streaming in
        # starts after the LAST resident chunk of the first entry that has trailing room (holes left by a partial stream out are not
        # refilled); streaming out starts at the first resident chunk.
        loaded = self.loaded[tier]
        for entry_index in range((len(loaded) + 31) // 32):
            entry = 0
            for bit in range(32):
                index = entry_index * 32 + bit
                if index < len(loaded) and loaded[index]:
                    entry |= 0x80000000 >> bit
            if stream_in:
                num_pending = 32 if entry == 0 else (entry & -entry).bit_length() - 1      # count_trailing_zeros
                if num_pending != 0:
                    return entry_index * 32 + (32 - num_pending)
            else:
                num_pending = 32 - entry.bit_length()                                       # count_leading_zeros
                if num_pending != 32:
                    return entry_index * 32 + num_pending
        return None

    def _apply(self, tier, first, last, stream_in):
        tier_index = tier - 1
        source = self.source[tier]
        for chunk_index in range(first, last + 1):
            _size, offset = (int(v) for v in self.chunks[tier][chunk_index])
            index, chunk_size, num_segments = (int(v) for v in _u32(source.tobytes(), offset, 3))
            assert index == chunk_index
            if stream_in:
                self.resident[tier][offset: offset + chunk_size] = source[offset: offset + chunk_size]
            segment_headers = _u32(source.tobytes(), offset + 12, 5 * num_segments).reshape(-1, 5)
            for _clip_hash, sample_indices, samples_offset, _clip_header_offset, segment_header_offset in segment_headers:
                value = ((int(samples_offset) << 32) | int(sample_indices)) if stream_in else 0
                at = int(segment_header_offset) + 8 * tier_index
                self.runtime_headers[at: at + 8] = np.frombuffer(np.uint64(value).tobytes(), dtype=np.uint8)
            self.loaded[tier][chunk_index] = stream_in

    def stream_in(self, tier, num_chunks=0xFFFFFFFF):
        """Returns the number of chunks streamed in (0 = database_stream_request_result::done)."""
        return self._stream(tier, num_chunks, True)

    def stream_out(self, tier, num_chunks=0xFFFFFFFF):
        return self._stream(tier, num_chunks, False)

    def _stream(self, tier, num_chunks, stream_in):
        total = self.num_chunks[tier]
        num_chunks = min(num_chunks, total)
        if total == 0:
            return 0
        first = self._first_chunk(tier, stream_in)
        if first is None or first >= total:
            return 0
        # database.impl.h:490-497,571-578 in the reference's unsigned 64 bit arithmetic: a request for 0 chunks wraps when the first
        # candidate is chunk 0 (the whole tier moves) and selects nothing otherwise
        last64 = (first + num_chunks - 1) & 0xFFFFFFFFFFFFFFFF
        last = total - 1 if last64 >= total else last64
        if last - first + 1 == 0:
            return 0
        self._apply(tier, first, last, stream_in)
        return last - first + 1

    # ---- decode ----------------------------------------------------------------------------------------------
    def binding(self):
        """The aclo_database view of the current state (arrays stay owned by this object and are updated in place)."""
        if self._binding is None:
            self._binding = bindings.Database()
            self._binding.clip_segment_headers = self.runtime_headers.ctypes.data
            self._binding.bulk_data[0] = self.resident[TIER_MEDIUM].ctypes.data
            self._binding.bulk_data[1] = self.resident[TIER_LOW].ctypes.data
        return self._binding

    def options(self, **overrides):
        options = bindings.default_options(**overrides)
        options.database = ctypes.pointer(self.binding())
        return options

    def decompress_tracks(self, clip_blob, sample_time, rounding=bindings.ROUND_NONE, **overrides):
        return bindings.oracle_decompress_tracks(clip_blob, sample_time, rounding, self.options(**overrides))

    def decompress_track(self, clip_blob, sample_time, track_index, rounding=bindings.ROUND_NONE, **overrides):
        return bindings.oracle_decompress_track(clip_blob, sample_time, track_index, rounding, self.options(**overrides))

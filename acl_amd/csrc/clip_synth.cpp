// clip_synth.cpp -- see clip_synth.h. Host-only; built into libaclsynth.so.
#include "clip_synth.h"
#include "acl_format.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace
{
	using namespace aclhip;

	// splitmix64 -> uniform floats; deterministic across platforms
	struct rng_t
	{
		uint64_t state;
		explicit rng_t(uint64_t seed) : state(seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull) {}
		uint64_t next()
		{
			uint64_t z = (state += 0x9E3779B97F4A7C15ull);
			z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
			z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
			return z ^ (z >> 31);
		}
		double uniform() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }		// [0, 1)
		double range(double lo, double hi) { return lo + (hi - lo) * uniform(); }
		uint32_t below(uint32_t n) { return uint32_t(uniform() * n); }
	};

	struct float3 { float x, y, z; };

	enum sub_track_kind { k_rotation = 0, k_translation = 1, k_scale = 2 };

	struct sub_track_t
	{
		uint32_t cls = k_sub_track_default;
		std::vector<float3> raw;			// num_samples (animated) or 1 (constant)
		float3 clip_min = { 0, 0, 0 };
		float3 clip_extent = { 0, 0, 0 };
		std::vector<float3> normalized;		// clip normalized, animated only
	};

	struct segment_sub_track_t
	{
		uint32_t num_bits = 0;				// 0 = constant in segment, 32 = raw, else 1..23
		uint8_t range_min[3] = { 0, 0, 0 };	// u8 segment range
		uint8_t range_extent[3] = { 0, 0, 0 };
		uint16_t constant_sample[3] = { 0, 0, 0 };	// width 0: 16 bit clip normalized sample
		std::vector<uint32_t> quantized;	// 3 per sample of the segment (not for raw / width 0)
	};

	struct segment_t
	{
		uint32_t start = 0;
		uint32_t num_samples = 0;
		uint32_t sample_indices = 0xFFFFFFFFu;	// stored keyframes, MSB = sample 0
		uint32_t num_stored = 0;
		std::vector<segment_sub_track_t> rotations, translations, scales;
		uint32_t rotation_bits = 0, translation_bits = 0, scale_bits = 0;
		uint32_t data_offset = 0;				// relative to transform header
	};

	inline float& comp(float3& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : v.z); }
	inline float comp(const float3& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : v.z); }

	// Big-endian MSB-first bit writer (core/memory_utils.h:295-335 semantics for a zeroed destination)
	struct bit_writer
	{
		uint8_t* base;
		uint64_t bit_offset = 0;
		explicit bit_writer(uint8_t* base_) : base(base_) {}
		void write(uint32_t value, uint32_t num_bits)
		{
			for (uint32_t i = 0; i < num_bits; ++i)
			{
				const uint32_t bit = (value >> (num_bits - 1 - i)) & 1u;
				if (bit)
					base[bit_offset >> 3] |= uint8_t(0x80u >> (bit_offset & 7));
				bit_offset++;
			}
		}
	};

	inline uint32_t float_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

	void split_samples_per_segment(uint32_t num_samples, uint32_t ideal, uint32_t max_samples, std::vector<uint32_t>& out)
	{
		// compression/impl/segment.transform.h:65-128
		out.clear();
		if (num_samples <= max_samples)
		{
			out.push_back(num_samples);
			return;
		}

		const uint32_t num_estimated = (num_samples + ideal - 1) / ideal;
		const uint32_t max_num = num_estimated * ideal;
		out.assign(num_estimated, ideal);
		const uint32_t last = num_estimated - 1;
		const uint32_t num_last = ideal - (max_num - num_samples);
		out[last] = num_last;

		const uint32_t slack_per_segment = max_samples - ideal;
		const uint32_t total_slack = last * slack_per_segment;
		if (total_slack >= num_last)
		{
			while (out[last] != 0)
			{
				for (uint32_t i = 0; i < last; ++i)
				{
					if (out[last] == 0)
						break;
					out[i]++;
					out[last]--;
				}
			}
			out.pop_back();
		}
	}

	struct track_t { sub_track_t sub[3]; };

	void generate_raw(rng_t& rng, const aclsynth_spec& spec, std::vector<track_t>& tracks)
	{
		const uint32_t n = spec.num_samples;
		const double duration = n > 1 ? double(n - 1) / spec.sample_rate : 1.0;

		for (uint32_t t = 0; t < spec.num_tracks; ++t)
		{
			for (int kind = 0; kind < 3; ++kind)
			{
				sub_track_t& st = tracks[t].sub[kind];
				if (kind == k_scale && !spec.has_scale)
				{
					st.cls = k_sub_track_default;
					continue;
				}

				const float p_default = kind == k_rotation ? spec.rotation_default : (kind == k_translation ? spec.translation_default : spec.scale_default);
				const float p_constant = kind == k_rotation ? spec.rotation_constant : (kind == k_translation ? spec.translation_constant : spec.scale_constant);
				const double u = rng.uniform();
				st.cls = u < p_default ? k_sub_track_default : (u < double(p_default) + p_constant ? k_sub_track_constant : k_sub_track_animated);
				if (n <= 1 && st.cls == k_sub_track_animated)
					st.cls = k_sub_track_constant;	// a single sample cannot be animated

				if (st.cls == k_sub_track_default)
					continue;

				const uint32_t count = st.cls == k_sub_track_animated ? n : 1;
				st.raw.resize(count);

				// A handful of sinusoids with random phase gives smooth, band-limited motion
				double freq[3][3], phase[3][3], amp[3][3];
				for (int c = 0; c < 3; ++c)
					for (int k = 0; k < 3; ++k)
					{
						freq[c][k] = rng.range(0.2, 2.5) * (k + 1) * 6.283185307179586 / duration;
						phase[c][k] = rng.range(0.0, 6.283185307179586);
						amp[c][k] = rng.range(0.2, 1.0) / (k + 1);
					}

				if (kind == k_rotation)
				{
					double base[4];
					double len = 0.0;
					for (int c = 0; c < 4; ++c) { base[c] = rng.range(-1.0, 1.0); len += base[c] * base[c]; }
					len = std::sqrt(len > 1e-12 ? len : 1.0);
					for (int c = 0; c < 4; ++c) base[c] /= len;
					const double swing = rng.range(0.05, 0.45);

					for (uint32_t s = 0; s < count; ++s)
					{
						const double time = double(s) / spec.sample_rate;
						double q[4] = { base[0], base[1], base[2], base[3] };
						for (int c = 0; c < 3; ++c)
							for (int k = 0; k < 3; ++k)
								q[c] += swing * amp[c][k] * std::sin(freq[c][k] * time + phase[c][k]);
						double l = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
						if (l < 1e-9) { q[3] = 1.0; l = 1.0; }
						const double sign = q[3] < 0.0 ? -1.0 : 1.0;	// W is dropped: keep it positive (math/quat_packing.h:58-62)
						st.raw[s] = { float(sign * q[0] / l), float(sign * q[1] / l), float(sign * q[2] / l) };
					}
				}
				else
				{
					const double extent = kind == k_translation ? double(spec.translation_extent) : 0.25;
					const double centre = kind == k_translation ? 0.0 : 1.0;
					double base[3];
					for (int c = 0; c < 3; ++c) base[c] = centre + rng.range(-0.6, 0.6) * extent;
					const double swing = rng.range(0.05, 0.35) * extent;
					// mirrored rigs: whole components of a scale sub-track are negative (drawn only when asked for: older seeds keep their clips)
					double mirror[3] = { 1.0, 1.0, 1.0 };
					if (kind == k_scale && spec.mirrored_scale_fraction > 0.0f)
						for (int c = 0; c < 3; ++c)
							mirror[c] = rng.uniform() < double(spec.mirrored_scale_fraction) ? -1.0 : 1.0;

					for (uint32_t s = 0; s < count; ++s)
					{
						const double time = double(s) / spec.sample_rate;
						double v[3];
						for (int c = 0; c < 3; ++c)
						{
							v[c] = base[c];
							for (int k = 0; k < 3; ++k)
								v[c] += swing * amp[c][k] * 0.5 * std::sin(freq[c][k] * time + phase[c][k]);
							v[c] = std::min(std::max(v[c], centre - extent), centre + extent);
						}
						st.raw[s] = { float(mirror[0] * v[0]), float(mirror[1] * v[1]), float(mirror[2] * v[2]) };
					}
				}
			}
		}
	}

	void clip_normalize(sub_track_t& st)
	{
		// compression/impl/normalize.transform.h:200-262 (min/extent over the clip, n = (v - min) / extent clamped to 1)
		float3 mn = st.raw[0], mx = st.raw[0];
		for (const float3& v : st.raw)
			for (int c = 0; c < 3; ++c)
			{
				comp(mn, c) = std::min(comp(mn, c), comp(v, c));
				comp(mx, c) = std::max(comp(mx, c), comp(v, c));
			}
		st.clip_min = mn;
		for (int c = 0; c < 3; ++c)
			comp(st.clip_extent, c) = comp(mx, c) - comp(mn, c);

		st.normalized.resize(st.raw.size());
		for (size_t s = 0; s < st.raw.size(); ++s)
			for (int c = 0; c < 3; ++c)
			{
				const float extent = comp(st.clip_extent, c);
				float nv = extent < 1.0e-9f ? 0.0f : (comp(st.raw[s], c) - comp(mn, c)) / extent;
				nv = std::min(std::max(nv, 0.0f), 1.0f);
				comp(st.normalized[s], c) = nv;
			}
	}

	uint32_t pick_num_bits(rng_t& rng, const aclsynth_spec& spec, bool multi_segment)
	{
		const double u = rng.uniform();
		if (u < spec.raw_fraction)
			return 32;
		if (multi_segment && u < double(spec.raw_fraction) + spec.width0_fraction)
			return 0;

		uint32_t lo = spec.min_bits, hi = spec.max_bits;
		const uint32_t floor_bits = spec.version <= k_version_v02_01_99 ? 3u : 1u;		// core/impl/variable_bit_rates.h:42-45
		const uint32_t ceil_bits = spec.version <= k_version_v02_01_99 ? 19u : 23u;
		lo = std::min(std::max(lo, floor_bits), ceil_bits);
		hi = std::min(std::max(hi, lo), ceil_bits);
		return lo + rng.below(hi - lo + 1);
	}

	void quantize_segment_sub_track(rng_t& rng, const aclsynth_spec& spec, const sub_track_t& st, const segment_t& seg, bool multi_segment, segment_sub_track_t& out)
	{
		out.num_bits = pick_num_bits(rng, spec, multi_segment);

		// Segment range, u8 with outward padding (normalize.transform.h:133-171)
		float seg_min[3] = { 0, 0, 0 }, seg_ext[3] = { 1, 1, 1 };
		if (multi_segment)
		{
			for (int c = 0; c < 3; ++c)
			{
				float mn = 1.0f, mx = 0.0f;
				for (uint32_t s = 0; s < seg.num_samples; ++s)
				{
					const float v = comp(st.normalized[seg.start + s], c);
					mn = std::min(mn, v);
					mx = std::max(mx, v);
				}

				int32_t min_q = int32_t(std::floor(mn * 255.0f));
				min_q = std::min(std::max(min_q, 0), 255);
				while (min_q > 0 && float(min_q) * (1.0f / 255.0f) > mn)
					min_q--;

				const float min_f = float(min_q) * (1.0f / 255.0f);
				int32_t ext_q = int32_t(std::ceil((mx - min_f) * 255.0f));
				ext_q = std::min(std::max(ext_q, 0), 255);
				while (ext_q < 255 && min_f + float(ext_q) * (1.0f / 255.0f) < mx)
					ext_q++;

				out.range_min[c] = uint8_t(min_q);
				out.range_extent[c] = uint8_t(ext_q);
				seg_min[c] = min_f;
				seg_ext[c] = float(ext_q) * (1.0f / 255.0f);
			}
		}

		if (out.num_bits == 0)
		{
			// First clip-normalized sample of the segment on 16 bits (quantize.transform.h:383-394)
			for (int c = 0; c < 3; ++c)
				out.constant_sample[c] = uint16_t(std::floor(comp(st.normalized[seg.start], c) * 65535.0f + 0.5f));
			return;
		}

		if (out.num_bits == 32)
			return;		// raw samples are written straight from st.raw

		const float max_value = float((1u << out.num_bits) - 1);
		out.quantized.resize(size_t(seg.num_samples) * 3);
		for (uint32_t s = 0; s < seg.num_samples; ++s)
			for (int c = 0; c < 3; ++c)
			{
				float nv = comp(st.normalized[seg.start + s], c);
				if (multi_segment)
				{
					nv = seg_ext[c] > 0.0f ? (nv - seg_min[c]) / seg_ext[c] : 0.0f;
					nv = std::min(std::max(nv, 0.0f), 1.0f);
				}
				// math/scalar_packing.h:42-48: round half away from zero of value * max
				out.quantized[size_t(s) * 3 + c] = uint32_t(std::floor(nv * max_value + 0.5f));
			}
	}

	// Value a decoder must produce AT a stored keyframe, from the quantized data, in double precision
	void expected_value(const sub_track_t& st, const segment_t& seg, const segment_sub_track_t& sst, bool multi_segment, uint32_t segment_sample, double out[3])
	{
		for (int c = 0; c < 3; ++c)
		{
			double v;
			if (sst.num_bits == 32)
			{
				out[c] = double(comp(st.raw[seg.start + segment_sample], c));
				continue;
			}
			else if (sst.num_bits == 0)
				v = double(sst.constant_sample[c]) / 65535.0;
			else
			{
				v = double(sst.quantized[size_t(segment_sample) * 3 + c]) / double((1u << sst.num_bits) - 1);
				if (multi_segment)
					v = v * (double(sst.range_extent[c]) / 255.0) + (double(sst.range_min[c]) / 255.0);
			}
			out[c] = v * double(comp(st.clip_extent, c)) + double(comp(st.clip_min, c));
		}
	}
}

extern "C" void aclsynth_default_spec(aclsynth_spec* spec)
{
	std::memset(spec, 0, sizeof(*spec));
	spec->seed = 1;
	spec->num_tracks = 100;
	spec->num_samples = 301;
	spec->sample_rate = 30.0f;
	spec->version = k_version_latest;
	spec->has_scale = 0;
	spec->default_scale = 1;
	spec->wrap = 0;
	spec->strip_keyframes = 0;
	spec->strip_fraction = 0.3f;
	spec->rotation_default = 0.02f;
	spec->rotation_constant = 0.62f;
	spec->translation_default = 0.02f;
	spec->translation_constant = 0.95f;
	spec->scale_default = 0.75f;
	spec->scale_constant = 0.05f;
	spec->min_bits = 8;
	spec->max_bits = 16;
	spec->width0_fraction = 0.03f;
	spec->raw_fraction = 0.01f;
	spec->translation_extent = 2.0f;
	spec->ideal_segment_samples = 16;
	spec->max_segment_samples = 31;
	spec->mirrored_scale_fraction = 0.0f;
}

extern "C" uint32_t aclsynth_build_clip(const aclsynth_spec* spec_, void* out, uint32_t capacity,
	float* expected_keyframes, uint8_t* stored_keyframes, float* raw_keyframes)
{
	if (spec_ == nullptr)
		return 0;

	const aclsynth_spec& spec = *spec_;
	if (spec.version < k_version_first || spec.version > k_version_latest)
		return 0;
	if (spec.num_tracks != 0 && (spec.num_samples == 0 || !(spec.sample_rate > 0.0f)))
		return 0;
	if (spec.ideal_segment_samples == 0 || spec.max_segment_samples < spec.ideal_segment_samples || spec.max_segment_samples > 32)
		return 0;
	if (spec.strip_keyframes && spec.version < k_version_v02_01_99)
		return 0;	// keyframe stripping is an ACL 2.1 feature

	rng_t rng(spec.seed);

	const uint32_t num_tracks = spec.num_tracks;
	const uint32_t num_samples = num_tracks != 0 ? spec.num_samples : 0;
	const bool has_scale = spec.has_scale != 0;

	std::vector<track_t> tracks(num_tracks);
	generate_raw(rng, spec, tracks);

	// Index animated / constant sub-tracks in track order per kind
	std::vector<uint32_t> animated[3], constant[3];
	for (uint32_t t = 0; t < num_tracks; ++t)
		for (int kind = 0; kind < 3; ++kind)
		{
			sub_track_t& st = tracks[t].sub[kind];
			if (st.cls == k_sub_track_animated) { clip_normalize(st); animated[kind].push_back(t); }
			else if (st.cls == k_sub_track_constant) constant[kind].push_back(t);
		}

	// Segments
	std::vector<uint32_t> samples_per_segment;
	if (num_samples != 0)
		split_samples_per_segment(num_samples, spec.ideal_segment_samples, spec.max_segment_samples, samples_per_segment);
	else
		samples_per_segment.push_back(0);

	const uint32_t num_segments = uint32_t(samples_per_segment.size());
	const bool multi_segment = num_segments > 1;
	const bool stripped = spec.strip_keyframes != 0;

	std::vector<segment_t> segments(num_segments);
	{
		uint32_t start = 0;
		for (uint32_t i = 0; i < num_segments; ++i)
		{
			segment_t& seg = segments[i];
			seg.start = start;
			seg.num_samples = samples_per_segment[i];
			start += seg.num_samples;

			// Stored keyframes: first and last of a segment always stay
			uint32_t indices = 0;
			for (uint32_t s = 0; s < seg.num_samples; ++s)
			{
				const bool boundary = s == 0 || s + 1 == seg.num_samples;
				const bool keep = !stripped || boundary || rng.uniform() >= spec.strip_fraction;
				if (keep)
					indices |= 0x80000000u >> s;
			}
			seg.sample_indices = indices;
			seg.num_stored = uint32_t(__builtin_popcount(indices));

			for (int kind = 0; kind < 3; ++kind)
			{
				std::vector<segment_sub_track_t>& list = kind == k_rotation ? seg.rotations : (kind == k_translation ? seg.translations : seg.scales);
				list.resize(animated[kind].size());
				uint32_t bits = 0;
				for (size_t a = 0; a < animated[kind].size(); ++a)
				{
					quantize_segment_sub_track(rng, spec, tracks[animated[kind][a]].sub[kind], seg, multi_segment, list[a]);
					bits += list[a].num_bits * 3;
				}
				(kind == k_rotation ? seg.rotation_bits : (kind == k_translation ? seg.translation_bits : seg.scale_bits)) = bits;
			}
		}
	}

	// ---- sizes and offsets (compress.transform.impl.h:290-360, 431-460) ----
	const uint32_t num_animated_rotations = uint32_t(animated[k_rotation].size());
	const uint32_t num_animated_translations = uint32_t(animated[k_translation].size());
	const uint32_t num_animated_scales = uint32_t(animated[k_scale].size());
	const uint32_t num_rotations_padded = align_to_u32(num_animated_rotations, 4);
	const uint32_t num_animated_variable = num_rotations_padded + num_animated_translations + num_animated_scales;

	const uint32_t num_sub_track_entries = (num_tracks + 15) / 16;
	const uint32_t packed_types_size = num_sub_track_entries * (has_scale ? 3u : 2u) * 4u;
	const uint32_t constant_data_size = uint32_t(constant[0].size() + constant[1].size() + constant[2].size()) * 12u;
	const uint32_t clip_range_data_size = (num_animated_rotations + num_animated_translations + num_animated_scales) * 24u;
	const uint32_t segment_start_indices_size = multi_segment ? 4u * (num_segments + 1) : 0u;
	const uint32_t segment_header_size = stripped ? uint32_t(sizeof(stripped_segment_header)) : uint32_t(sizeof(segment_header));

	// all offsets below are relative to the transform_tracks_header
	const uint32_t segment_headers_offset = align_to_u32(k_segment_start_indices_offset + segment_start_indices_size, 4);
	const uint32_t sub_track_types_offset = align_to_u32(segment_headers_offset + segment_header_size * num_segments, 4);
	const uint32_t constant_track_data_offset = align_to_u32(sub_track_types_offset + packed_types_size, 4);
	const uint32_t clip_range_data_offset = align_to_u32(constant_track_data_offset + constant_data_size, 4);

	const uint32_t format_per_track_size = num_animated_variable;
	const uint32_t segment_range_size = multi_segment ? 6u * num_animated_variable : 0u;

	uint32_t cursor = k_transform_header_offset + clip_range_data_offset + clip_range_data_size;	// absolute, per compress.transform.impl.h:321-360
	for (segment_t& seg : segments)
	{
		seg.data_offset = cursor - k_transform_header_offset;
		const uint32_t pose_bits = seg.rotation_bits + seg.translation_bits + seg.scale_bits;
		const uint32_t animated_data_size = uint32_t((uint64_t(pose_bits) * seg.num_stored + 7) / 8);

		cursor += format_per_track_size;
		cursor = align_to_u32(cursor, 2);
		cursor += segment_range_size;
		cursor = align_to_u32(cursor, 4);
		cursor += animated_data_size;
	}

	const uint32_t total_size = cursor + 15;	// padding for unaligned 16 byte loads (compress.transform.impl.h:396)

	// ---- optional side outputs ----
	if (stored_keyframes != nullptr)
		for (const segment_t& seg : segments)
			for (uint32_t s = 0; s < seg.num_samples; ++s)
				stored_keyframes[seg.start + s] = (seg.sample_indices & (0x80000000u >> s)) != 0 ? 1 : 0;

	auto fill_pose_defaults = [&](float* pose)
	{
		for (uint32_t t = 0; t < num_tracks; ++t)
		{
			float* qvv = pose + size_t(t) * 12;
			qvv[0] = 0; qvv[1] = 0; qvv[2] = 0; qvv[3] = 1;
			qvv[4] = 0; qvv[5] = 0; qvv[6] = 0; qvv[7] = 0;
			const float ds = float(spec.default_scale);
			qvv[8] = ds; qvv[9] = ds; qvv[10] = ds; qvv[11] = 0;
		}
	};

	auto store_sub_track = [](float* qvv, int kind, const double v[3])
	{
		if (kind == k_rotation)
		{
			// W is rebuilt from xyz and the result is normalized, like a decoder that always lerps then normalizes
			// (quantization can push xyz slightly outside the unit ball, |w2| then makes the quaternion non-unit)
			const double w = std::sqrt(std::fabs(((1.0 - v[0] * v[0]) - v[1] * v[1]) - v[2] * v[2]));
			const double inv_len = 1.0 / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + w * w);
			qvv[0] = float(v[0] * inv_len); qvv[1] = float(v[1] * inv_len); qvv[2] = float(v[2] * inv_len); qvv[3] = float(w * inv_len);
		}
		else
		{
			float* dst = qvv + (kind == k_translation ? 4 : 8);
			dst[0] = float(v[0]); dst[1] = float(v[1]); dst[2] = float(v[2]); dst[3] = 0.0f;
		}
	};

	if (expected_keyframes != nullptr || raw_keyframes != nullptr)
	{
		for (uint32_t si = 0; si < num_segments; ++si)
		{
			const segment_t& seg = segments[si];
			for (uint32_t s = 0; s < seg.num_samples; ++s)
			{
				const uint32_t sample = seg.start + s;
				float* expected_pose = expected_keyframes != nullptr ? expected_keyframes + size_t(sample) * num_tracks * 12 : nullptr;
				float* raw_pose = raw_keyframes != nullptr ? raw_keyframes + size_t(sample) * num_tracks * 12 : nullptr;
				if (expected_pose != nullptr) fill_pose_defaults(expected_pose);
				if (raw_pose != nullptr) fill_pose_defaults(raw_pose);

				for (int kind = 0; kind < 3; ++kind)
				{
					for (uint32_t t : constant[kind])
					{
						const float3& v = tracks[t].sub[kind].raw[0];
						const double dv[3] = { v.x, v.y, v.z };
						if (expected_pose != nullptr) store_sub_track(expected_pose + size_t(t) * 12, kind, dv);
						if (raw_pose != nullptr) store_sub_track(raw_pose + size_t(t) * 12, kind, dv);
					}

					const std::vector<segment_sub_track_t>& list = kind == k_rotation ? seg.rotations : (kind == k_translation ? seg.translations : seg.scales);
					for (size_t a = 0; a < animated[kind].size(); ++a)
					{
						const uint32_t t = animated[kind][a];
						const sub_track_t& st = tracks[t].sub[kind];
						if (expected_pose != nullptr)
						{
							double dv[3];
							expected_value(st, seg, list[a], multi_segment, s, dv);
							store_sub_track(expected_pose + size_t(t) * 12, kind, dv);
						}
						if (raw_pose != nullptr)
						{
							const float3& v = st.raw[sample];
							const double dv[3] = { v.x, v.y, v.z };
							store_sub_track(raw_pose + size_t(t) * 12, kind, dv);
						}
					}
				}
			}
		}
	}

	if (out == nullptr || capacity < total_size)
		return total_size;

	// ---- write the blob ----
	uint8_t* blob = static_cast<uint8_t*>(out);
	std::memset(blob, 0, total_size);

	raw_buffer_header* buffer_header = reinterpret_cast<raw_buffer_header*>(blob);
	tracks_header* header = reinterpret_cast<tracks_header*>(blob + k_tracks_header_offset);
	transform_tracks_header* transforms = reinterpret_cast<transform_tracks_header*>(blob + k_transform_header_offset);
	uint8_t* tbase = blob + k_transform_header_offset;

	header->tag = k_tag_compressed_tracks;
	header->version = uint16_t(spec.version);
	header->algorithm_type = k_algorithm_uniformly_sampled;
	header->track_type = k_track_type_qvvf;
	header->num_tracks = num_tracks;
	header->num_samples = num_samples;
	header->sample_rate = num_tracks != 0 ? spec.sample_rate : 0.0f;
	uint32_t misc = 0;
	misc |= has_scale ? 1u : 0u;
	misc |= (spec.default_scale & 1u) << 1;
	misc |= uint32_t(k_vector_vector3f_variable) << 2;
	misc |= uint32_t(k_vector_vector3f_variable) << 3;
	misc |= uint32_t(k_rotation_quatf_drop_w_variable) << 4;
	misc |= 1u << 9;									// defaults are the trivial identity values
	misc |= stripped ? (1u << 10) : 0u;
	misc |= (spec.wrap != 0 && spec.version > k_version_first) ? (1u << 30) : 0u;
	header->misc_packed = misc;

	transforms->num_segments = num_segments;
	transforms->num_animated_variable_sub_tracks = num_animated_variable;
	transforms->num_animated_rotation_sub_tracks = num_animated_rotations;
	transforms->num_animated_translation_sub_tracks = num_animated_translations;
	transforms->num_animated_scale_sub_tracks = num_animated_scales;
	transforms->num_constant_rotation_samples = uint32_t(constant[k_rotation].size());
	transforms->num_constant_translation_samples = uint32_t(constant[k_translation].size());
	transforms->num_constant_scale_samples = uint32_t(constant[k_scale].size());
	transforms->database_header_offset = k_invalid_offset;
	transforms->segment_headers_offset = segment_headers_offset;
	transforms->sub_track_types_offset = sub_track_types_offset;
	transforms->constant_track_data_offset = constant_track_data_offset;
	transforms->clip_range_data_offset = clip_range_data_offset;

	if (multi_segment)
	{
		uint32_t* start_indices = reinterpret_cast<uint32_t*>(tbase + k_segment_start_indices_offset);
		for (uint32_t i = 0; i < num_segments; ++i)
			start_indices[i] = segments[i].start;
		start_indices[num_segments] = 0xFFFFFFFFu;
	}

	for (uint32_t i = 0; i < num_segments; ++i)
	{
		const segment_t& seg = segments[i];
		segment_header* sh = reinterpret_cast<segment_header*>(tbase + segment_headers_offset + size_t(i) * segment_header_size);
		sh->animated_pose_bit_size = seg.rotation_bits + seg.translation_bits + seg.scale_bits;
		sh->animated_rotation_bit_size = seg.rotation_bits;
		sh->animated_translation_bit_size = seg.translation_bits;
		sh->segment_data = seg.data_offset;
		if (stripped)
			static_cast<stripped_segment_header*>(sh)->sample_indices = seg.sample_indices;
	}

	// packed sub-track types: rotations, translations, [scales]
	{
		uint32_t* types = reinterpret_cast<uint32_t*>(tbase + sub_track_types_offset);
		const int num_kinds = has_scale ? 3 : 2;
		for (int kind = 0; kind < num_kinds; ++kind)
			for (uint32_t t = 0; t < num_tracks; ++t)
				types[size_t(kind) * num_sub_track_entries + t / 16] |= tracks[t].sub[kind].cls << ((15 - (t % 16)) * 2);
	}

	// constant track data: rotations SOA in groups of 4 (last group unpadded), then translations, scales AOS
	{
		float* dst = reinterpret_cast<float*>(tbase + constant_track_data_offset);
		const std::vector<uint32_t>& rots = constant[k_rotation];
		for (size_t g = 0; g < rots.size(); g += 4)
		{
			const size_t group = std::min<size_t>(4, rots.size() - g);
			for (int c = 0; c < 3; ++c)
				for (size_t j = 0; j < group; ++j)
					*dst++ = comp(tracks[rots[g + j]].sub[k_rotation].raw[0], c);
		}
		for (int kind = k_translation; kind <= k_scale; ++kind)
			for (uint32_t t : constant[kind])
			{
				const float3& v = tracks[t].sub[kind].raw[0];
				*dst++ = v.x; *dst++ = v.y; *dst++ = v.z;
			}
	}

	// clip range data: rotations SOA per group (min xyz then extent xyz), then translations / scales AOS
	{
		float* dst = reinterpret_cast<float*>(tbase + clip_range_data_offset);
		const std::vector<uint32_t>& rots = animated[k_rotation];
		for (size_t g = 0; g < rots.size(); g += 4)
		{
			const size_t group = std::min<size_t>(4, rots.size() - g);
			for (int c = 0; c < 3; ++c)
				for (size_t j = 0; j < group; ++j)
					*dst++ = comp(tracks[rots[g + j]].sub[k_rotation].clip_min, c);
			for (int c = 0; c < 3; ++c)
				for (size_t j = 0; j < group; ++j)
					*dst++ = comp(tracks[rots[g + j]].sub[k_rotation].clip_extent, c);
		}
		for (int kind = k_translation; kind <= k_scale; ++kind)
			for (uint32_t t : animated[kind])
			{
				const sub_track_t& st = tracks[t].sub[kind];
				*dst++ = st.clip_min.x; *dst++ = st.clip_min.y; *dst++ = st.clip_min.z;
				*dst++ = st.clip_extent.x; *dst++ = st.clip_extent.y; *dst++ = st.clip_extent.z;
			}
	}

	// per segment data
	const uint32_t raw_num_bits_stored = spec.version >= k_version_v02_01_99_1 ? 31u : 32u;
	for (const segment_t& seg : segments)
	{
		uint8_t* format_per_track = tbase + seg.data_offset;
		uint8_t* range_data = blob + align_to_u32(k_transform_header_offset + seg.data_offset + format_per_track_size, 2);
		uint8_t* animated_data = blob + align_to_u32(uint32_t(range_data - blob) + segment_range_size, 4);

		// format per track: rotations (padded to 4), translations, scales
		{
			uint8_t* dst = format_per_track;
			for (size_t a = 0; a < seg.rotations.size(); ++a)
				dst[a] = uint8_t(seg.rotations[a].num_bits == 32 ? raw_num_bits_stored : seg.rotations[a].num_bits);
			dst += num_rotations_padded;
			for (const segment_sub_track_t& sst : seg.translations)
				*dst++ = uint8_t(sst.num_bits == 32 ? raw_num_bits_stored : sst.num_bits);
			for (const segment_sub_track_t& sst : seg.scales)
				*dst++ = uint8_t(sst.num_bits == 32 ? raw_num_bits_stored : sst.num_bits);
		}

		if (multi_segment)
		{
			// rotations: 24 byte groups, SOA; a width 0 sub-track stores its 16 bit sample as hi/lo byte pairs
			uint8_t* dst = range_data;
			for (size_t a = 0; a < seg.rotations.size(); ++a)
			{
				const segment_sub_track_t& sst = seg.rotations[a];
				uint8_t* group = dst + (a / 4) * 24;
				const size_t lane = a % 4;
				if (sst.num_bits == 0)
				{
					group[lane + 0] = uint8_t(sst.constant_sample[0] >> 8);
					group[lane + 4] = uint8_t(sst.constant_sample[0] & 0xFF);
					group[lane + 8] = uint8_t(sst.constant_sample[1] >> 8);
					group[lane + 12] = uint8_t(sst.constant_sample[1] & 0xFF);
					group[lane + 16] = uint8_t(sst.constant_sample[2] >> 8);
					group[lane + 20] = uint8_t(sst.constant_sample[2] & 0xFF);
				}
				else
				{
					// raw sub-tracks still carry (unused) range bytes
					group[lane + 0] = sst.range_min[0];
					group[lane + 4] = sst.range_min[1];
					group[lane + 8] = sst.range_min[2];
					group[lane + 12] = sst.range_extent[0];
					group[lane + 16] = sst.range_extent[1];
					group[lane + 20] = sst.range_extent[2];
				}
			}
			dst += size_t(num_rotations_padded) * 6;

			for (int kind = k_translation; kind <= k_scale; ++kind)
			{
				const std::vector<segment_sub_track_t>& list = kind == k_translation ? seg.translations : seg.scales;
				for (const segment_sub_track_t& sst : list)
				{
					if (sst.num_bits == 0)
					{
						// u16 x, y, z little endian (write_range_data.h:284-289)
						for (int c = 0; c < 3; ++c)
						{
							dst[c * 2 + 0] = uint8_t(sst.constant_sample[c] & 0xFF);
							dst[c * 2 + 1] = uint8_t(sst.constant_sample[c] >> 8);
						}
					}
					else
					{
						dst[0] = sst.range_min[0]; dst[1] = sst.range_min[1]; dst[2] = sst.range_min[2];
						dst[3] = sst.range_extent[0]; dst[4] = sst.range_extent[1]; dst[5] = sst.range_extent[2];
					}
					dst += 6;
				}
			}
		}

		// animated data: stored keyframes back to back, NOT byte aligned between keyframes
		{
			bit_writer writer(animated_data);
			for (uint32_t s = 0; s < seg.num_samples; ++s)
			{
				if ((seg.sample_indices & (0x80000000u >> s)) == 0)
					continue;	// stripped

				for (int kind = 0; kind < 3; ++kind)
				{
					const std::vector<segment_sub_track_t>& list = kind == k_rotation ? seg.rotations : (kind == k_translation ? seg.translations : seg.scales);
					for (size_t a = 0; a < list.size(); ++a)
					{
						const segment_sub_track_t& sst = list[a];
						if (sst.num_bits == 0)
							continue;

						if (sst.num_bits == 32)
						{
							const float3& v = tracks[animated[kind][a]].sub[kind].raw[seg.start + s];
							writer.write(float_bits(v.x), 32);
							writer.write(float_bits(v.y), 32);
							writer.write(float_bits(v.z), 32);
						}
						else
						{
							writer.write(sst.quantized[size_t(s) * 3 + 0], sst.num_bits);
							writer.write(sst.quantized[size_t(s) * 3 + 1], sst.num_bits);
							writer.write(sst.quantized[size_t(s) * 3 + 2], sst.num_bits);
						}
					}
				}
			}
		}
	}

	buffer_header->size = total_size;
	buffer_header->hash = hash32(blob + sizeof(raw_buffer_header), total_size - sizeof(raw_buffer_header));
	return total_size;
}

// clip_synth.h -- seeded synthetic clip writer: emits legal ACL `compressed_tracks` blobs (qvvf,
// quatf_drop_w_variable + vector3f_variable) so that benchmarks and tests have CMU-shaped inputs
// without the reference compressor. Host-only C++, no GPU dependency.
//
// The byte layout written here follows the reference's writers
// (/root/reference/includes/acl/compression/impl/write_stream_data.h:199-531, write_range_data.h:79-341,
//  write_segment_data.h:48-182, write_sub_track_types.h:43-163, compress.transform.impl.h:290-530);
// values are produced by a small range-reduce + quantize pipeline in the spirit of
// normalize.transform.h:133-262 and quantize.transform.h:383-408.
#pragma once

#include <stdint.h>

#if defined(__cplusplus)
extern "C" {
#endif

typedef struct aclsynth_spec
{
	uint32_t seed;
	uint32_t num_tracks;
	uint32_t num_samples;
	float sample_rate;

	uint32_t version;				// 7 (v02_00_00) .. 10 (v02_01_00, default)
	uint32_t has_scale;				// 0 / 1
	uint32_t default_scale;			// 0 / 1 (the legacy default scale bit)
	uint32_t wrap;					// 1 = clip is wrap optimized (looping)
	uint32_t strip_keyframes;		// 1 = has_stripped_keyframes, a random subset of keyframes is removed
	float strip_fraction;			// probability an interior keyframe of a segment is removed

	// sub-track class mix: probability of default / constant, remainder is animated
	float rotation_default, rotation_constant;
	float translation_default, translation_constant;
	float scale_default, scale_constant;

	// per (segment, animated sub-track) bit width distribution
	uint32_t min_bits, max_bits;	// uniform in [min_bits, max_bits]
	float width0_fraction;			// constant-in-segment (width 0); only honoured when there is more than one segment
	float raw_fraction;				// full precision (raw) rate

	float translation_extent;		// translations live in [-extent, extent]
	uint32_t ideal_segment_samples;	// 16
	uint32_t max_segment_samples;	// 31
	float mirrored_scale_fraction;	// probability that a component of a constant / animated scale sub-track is NEGATIVE (mirrored rigs); 0 by default
} aclsynth_spec;

// Fills 'spec' with the CMU-shaped defaults of SURVEY.md section 8(d): 100 bones, 301 samples @ 30 Hz,
// rotations 2/62/36 % default/constant/animated, translations 2/95/3 %, widths 8..16, 3 % width-0, 1 % raw.
void aclsynth_default_spec(aclsynth_spec* spec);

// Writes the blob into 'out' (must be 16-byte aligned, 'capacity' bytes). Returns the blob size in bytes;
// when 'capacity' is too small (or 'out' is null) nothing is written and the required size is returned.
// Returns 0 on an invalid spec.
//
// Optional outputs (each may be null):
//   expected_keyframes  [num_samples][num_tracks][12] floats: the value every sub-track should decode to
//                       exactly AT each stored keyframe (qvv: rot xyzw, trans xyz0, scale xyz0), computed
//                       in double precision from the quantized data, independently of any decoder.
//                       Default sub-tracks hold identity / 0 / default_scale.
//   stored_keyframes    [num_samples] bytes: 1 when that keyframe is present in the blob (0 = stripped)
//   raw_keyframes       [num_samples][num_tracks][12] floats: the lossless source animation
uint32_t aclsynth_build_clip(const aclsynth_spec* spec, void* out, uint32_t capacity,
	float* expected_keyframes, uint8_t* stored_keyframes, float* raw_keyframes);

#if defined(__cplusplus)
}
#endif

#pragma once
#include <cstring>
#include <type_traits>
namespace rtm { namespace rtm_impl {
	template<class dest_type_t, class src_type_t>
	inline dest_type_t bit_cast(src_type_t input) noexcept
	{
		static_assert(sizeof(dest_type_t) == sizeof(src_type_t), "bit_cast needs same-size types");
		dest_type_t result;
		std::memcpy(&result, &input, sizeof(dest_type_t));
		return result;
	}
} }

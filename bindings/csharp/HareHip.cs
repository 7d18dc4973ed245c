// HareHip.cs -- P/Invoke surface of libhare_hip.so (include/hare_hip.h), for the drop-in
// Spatial_Partition subclasses in this folder.  Compile these files INTO Hare.csproj (net7.0;net48,
// AllowUnsafeBlocks not required) next to the reference sources; libhare_hip.so goes beside the
// assembly or on LD_LIBRARY_PATH.
//
// NOT COMPILED IN THE BUILD CONTAINER: no .NET toolchain exists there (DESIGN.md).  The native side
// pins the layouts with static_asserts (api.cpp), tests/test_abi_exports.py checks every export, and
// tests/test_csharp_binding.py parses THIS file against include/hare_hip.h: every DllImport (incl.
// EntryPoint aliases) must name an export, with the header's parameter count, scalar widths and
// pointer positions; every [StructLayout] struct must have the header's field sequence and size.
using System;
using System.Runtime.InteropServices;

namespace Hare
{
    namespace Geometry
    {
        /// <summary>Hare.Geometry.Ray as the native library reads it (Hare_Geometry_Primitives.cs:393-429).</summary>
        [StructLayout(LayoutKind.Sequential, Pack = 8)]
        public struct hare_ray
        {
            public double x, y, z, dx, dy, dz;
        }

        /// <summary>Hare.Geometry.X_Event as the native library writes it (Hare_Geometry_Primitives.cs:435-481). 56 bytes.</summary>
        [StructLayout(LayoutKind.Sequential, Pack = 8)]
        public struct hare_xevent
        {
            public double t, u, v, x, y, z;
            public int poly_id;
            public int hit;
        }

        /// <summary>Slim result record of a Voxel_Grid batch (HARE_SHOOT_SLIM_EVENTS; include/hare_hip.h): 16 bytes instead of 56.
        /// hit = 1: X_Event.t = t, X_Point = R.origin + R.direction * t (Hare_Geometry_Polygons.cs:652; same bits), u = v = 0.
        /// hit = 2: the ray started outside the grid and its origin was moved; use HareHip.hare_expand_events.</summary>
        [StructLayout(LayoutKind.Sequential, Pack = 8)]
        public struct hare_slim_event
        {
            public double t;
            public int poly_id;
            public int hit;
        }

        /// <summary>Slim result record of an Octree / KDTree batch (they return u, v; rays are never moved). 32 bytes.</summary>
        [StructLayout(LayoutKind.Sequential, Pack = 8)]
        public struct hare_slim_event_uv
        {
            public double t, u, v;
            public int poly_id;
            public int hit;
        }

        [StructLayout(LayoutKind.Sequential, Pack = 8)]
        public struct hare_counters
        {
            public ulong rays, hits, cells, entries, tests, r0, r1, r2;
        }

        [StructLayout(LayoutKind.Sequential, Pack = 8)]
        public struct hare_topology_desc
        {
            public int P;
            public int reserved;
            public IntPtr verts;    // P x 4 x 3 doubles
            public IntPtr nverts;   // P ints
            public IntPtr normals;  // P x 3 doubles
            public double min0, min1, min2;
            public double max0, max1, max2;
        }

        [StructLayout(LayoutKind.Sequential, Pack = 8)]
        public struct hare_voxel_info
        {
            public int ct, n_topos;
            public double obox_min0, obox_min1, obox_min2;
            public double obox_max0, obox_max1, obox_max2;
            public double box_dims0, box_dims1, box_dims2;
            public double voxel_dims0, voxel_dims1, voxel_dims2;
            public double char_step;
            public ulong total_items;
            public int built_on_device, reserved;
        }

        [StructLayout(LayoutKind.Sequential, Pack = 8)]
        public struct hare_tree_info
        {
            public int n_nodes, max_depth, max_polys, built_on_device;
            public ulong total_items;
        }

        internal static class HareHip
        {
            const string Lib = "hare_hip";   // libhare_hip.so

            public const int HARE_KIND_VOXEL = 0, HARE_KIND_OCTREE = 1, HARE_KIND_KDTREE = 2;
            public const uint HARE_SHOOT_WRITEBACK_ORIGIN = 1;
            public const uint HARE_SHOOT_RETIRED_RAYS = 8;   // device-resident bounce loop only (hare_shoot_device)
            public const uint HARE_SHOOT_COUNT_OWN = 64;     // measurement: the production kernel's counting build (its own cells / entries / pre-culls / tests)
            public const uint HARE_SHOOT_BOUNCE_LOOP = 32;   // hare_shoot_kernel_name only
            public const uint HARE_SHOOT_SLIM_EVENTS = 16;   // host-buffer batches: hare_slim_event records come back (16 B per ray, not 56)

            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern IntPtr hare_last_error();
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int hare_device_count(out int count);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int hare_scene_create([In] hare_topology_desc[] topos, int n_topos, int device, out IntPtr scene);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern void hare_scene_destroy(IntPtr scene);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int hare_voxel_build(IntPtr scene, int domain);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int hare_voxel_build_adaptive(IntPtr scene, int max_domain, int avg_polys);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int hare_octree_build(IntPtr scene, int max_depth, int max_polys);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int hare_kdtree_build(IntPtr scene, int max_depth, int max_polys);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int hare_voxel_get_info(IntPtr scene, out hare_voxel_info info);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_shoot_batch(IntPtr scene, int kind, int top_index, long n, [In, Out] hare_ray[] rays,
                                                      int[] excl1, int[] excl2, uint flags, [Out] hare_xevent[] ev, out hare_counters ctr);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern unsafe int hare_shoot_batch(IntPtr scene, int kind, int top_index, long n, hare_ray* rays,
                                                             int* excl1, int* excl2, uint flags, hare_xevent* ev, hare_counters* ctr);
            /// <summary>One batch over several scenes (one per device), contiguous shards (include/hare_hip.h).</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_shoot_batch_sharded([In] IntPtr[] scenes, int n_scenes, int kind, int top_index, long n,
                                                              [In, Out] hare_ray[] rays, int[] excl1, int[] excl2, uint flags,
                                                              [Out] hare_xevent[] ev, out hare_counters ctr);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern unsafe int hare_shoot_batch_sharded([In] IntPtr[] scenes, int n_scenes, int kind, int top_index, long n,
                                                                     hare_ray* rays, int* excl1, int* excl2, uint flags,
                                                                     hare_xevent* ev, hare_counters* ctr);
            /// <summary>Voxel_Grid batch with slim result records (flags must contain HARE_SHOOT_SLIM_EVENTS).</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl, EntryPoint = "hare_shoot_batch_sharded")]
            public static extern int hare_shoot_batch_sharded_slim([In] IntPtr[] scenes, int n_scenes, int kind, int top_index, long n,
                                                                   [In] hare_ray[] rays, int[] excl1, int[] excl2, uint flags,
                                                                   [Out] hare_slim_event[] ev, out hare_counters ctr);
            /// <summary>Slim records back to full X_Event records on the host, bit for bit (needs no GPU).</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_expand_events(IntPtr scene, int kind, long n, [In] hare_ray[] rays, [In] hare_slim_event[] slim, [Out] hare_xevent[] ev);
            /// <summary>The device-resident specular bounce loop from host buffers (include/hare_hip.h): `bounces` casts with a
            /// reflection about Normal(Poly_id) between them; events_all is bounces x n records, cast-major; any output may be null.</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_bounce_batch_sharded([In] IntPtr[] scenes, int n_scenes, int kind, int top_index, long n,
                                                               [In] hare_ray[] rays, int[] excl1, int[] excl2, int bounces, uint flags,
                                                               [Out] hare_xevent[] events_all, [Out] hare_xevent[] events_last,
                                                               out hare_counters ctr, [Out] hare_counters[] ctr_per_cast);
            /// <summary>Diagnostics / A-B switch of one scene ("voxel_kernel", "octree_kernel", "build_host", ...: include/hare_hip.h).</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl, CharSet = CharSet.Ansi)]
            public static extern int hare_scene_set_option(IntPtr scene, string name, long value);
            /// <summary>An option read back, or "voxel_tight_bytes" / "octree_scratch_bytes": device memory of the accelerators (include/hare_hip.h).</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl, CharSet = CharSet.Ansi)]
            public static extern int hare_scene_get_option(IntPtr scene, string name, out long value);
            /// <summary>Spatial_Partition.Shoot for ONE ray on the calling thread (host trace, no GPU round trip, lock-free):
            /// what the single-ray overrides call.  ray is updated like the reference moves R (AABB_Main.cs:254-257).</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_shoot_one(IntPtr scene, int kind, int top_index, ref hare_ray ray, int poly_origin1, int poly_origin2, out hare_xevent ev);
            /// <summary>Voxel_Inv[x,y,z,top] as CSR: cell = (x*ct + y)*ct + z; items may be null to size the buffer (cell_start[ct^3]).</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_voxel_get_lists(IntPtr scene, int top_index, [Out] uint[] cell_start, [Out] int[] items);
            /// <summary>Occlusion predicate (harness-defined): occluded[i] = closest hit exists and t &lt; tmax[i] (tmax null: any hit).</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_occluded_batch(IntPtr scene, int kind, int top_index, long n, [In, Out] hare_ray[] rays, int[] excl1, int[] excl2,
                                                         double[] tmax, uint flags, [Out] int[] occluded, [Out] hare_xevent[] events, out hare_counters ctr);
            /// <summary>The same over several scenes (one per device), contiguous shards; events may be null (flags only: the traversal
            /// of a ray ends as soon as its flag is decided, 4 bytes per ray come back instead of 56).</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_occluded_batch_sharded([In] IntPtr[] scenes, int n_scenes, int kind, int top_index, long n, [In] hare_ray[] rays,
                                                                 int[] excl1, int[] excl2, double[] tmax, uint flags, [Out] int[] occluded,
                                                                 [Out] hare_xevent[] events, out hare_counters ctr);
            /// <summary>Topology(Point[][]) ingest for hosts holding a raw polygon soup (include/hare_hip.h).</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_topology_ingest([In] double[] soup, [In] int[] nverts, int P, [Out] double[] verts_out,
                                                          [Out] int[] corner_vertex, [Out] double[] vertices_out, out int n_vertices);

            // ---- the rest of include/hare_hip.h, so that every export has a managed declaration (tests/test_csharp_binding.py checks
            // each one against the header: entry point, parameter count, scalar widths, pointer-ness, struct layouts)
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern IntPtr hare_version();
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern IntPtr hare_hip_runtime_path();
            /// <summary>Polygon ctor normals (Hare_Geometry_Polygons.cs:159-171) for hosts that bypass the managed Topology.</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_polygon_normals([In] double[] verts, [In] int[] nverts, int P, [Out] double[] normals_out);
            /// <summary>Finish_Topology bounds (Hare_Geometry_Topology.cs:148-167).</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_topology_bounds([In] double[] verts, [In] int[] nverts, int P, [Out] double[] min_out, [Out] double[] max_out);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int hare_octree_get_info(IntPtr scene, out hare_tree_info info);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_octree_get_nodes(IntPtr scene, [Out] double[] boxes, [Out] int[] first_child, [Out] int[] item_start,
                                                           [Out] int[] item_count, [Out] int[] items);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int hare_kdtree_get_info(IntPtr scene, out hare_tree_info info);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_kdtree_get_nodes(IntPtr scene, [Out] double[] boxes, [Out] double[] split, [Out] int[] axis, [Out] int[] left,
                                                           [Out] int[] right, [Out] int[] item_start, [Out] int[] item_count, [Out] int[] items);
            /// <summary>Name of the gfx950 kernel a batch of n rays would launch (profiles list kernels by this name).</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern IntPtr hare_shoot_kernel_name(IntPtr scene, int kind, int top_index, long n, uint flags);
            /// <summary>The bounce loop on ONE scene (hare_bounce_batch_sharded with a single scene does the same).</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_bounce_batch(IntPtr scene, int kind, int top_index, long n, [In] hare_ray[] rays, int[] excl1, int[] excl2,
                                                       int bounces, uint flags, [Out] hare_xevent[] events_all, [Out] hare_xevent[] events_last,
                                                       out hare_counters ctr, [Out] hare_counters[] ctr_per_cast);
            // Device-pointer entry points, for managed hosts that hold HIP allocations (e.g. through a HIP interop layer): every
            // pointer is a device address on the scene's device, `stream` a hipStream_t (IntPtr.Zero: the default stream).
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_shoot_device(IntPtr scene, int kind, int top_index, long n, IntPtr d_rays, IntPtr d_excl1, IntPtr d_excl2,
                                                       uint flags, IntPtr d_out, IntPtr d_counters, IntPtr stream);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_occluded_device(IntPtr scene, int kind, int top_index, long n, IntPtr d_rays, IntPtr d_excl1, IntPtr d_excl2,
                                                          IntPtr d_tmax, uint flags, IntPtr d_events, IntPtr d_occluded, IntPtr d_counters, IntPtr stream);
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_reflect_device(IntPtr scene, int top_index, long n, IntPtr d_rays, IntPtr d_events, IntPtr d_excl_out, IntPtr stream);
            /// <summary>The whole bounce loop on device buffers, stream-ordered (one launch for a Voxel_Grid where the pool kernel serves).
            /// d_rays is read and overwritten; d_work is 2 n int32 of scratch; d_events_all (bounces x n) and the counters may be null.</summary>
            [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
            public static extern int hare_bounce_device(IntPtr scene, int kind, int top_index, long n, IntPtr d_rays, IntPtr d_excl1, IntPtr d_excl2,
                                                        int bounces, uint flags, IntPtr d_work, IntPtr d_events_all, IntPtr d_events_last,
                                                        IntPtr d_counters, IntPtr d_counters_per_cast, IntPtr stream);

            public static void Check(int rc)
            {
                if (rc == 0) return;
                string msg = Marshal.PtrToStringAnsi(hare_last_error()) ?? "";
                if (rc == -6) throw new NotImplementedException(msg);          // as Topology does for > 4 corners
                throw new InvalidOperationException("hare_hip error " + rc + ": " + msg);
            }
        }
    }
}

// Gpu_Spatial_Partition.cs -- drop-in Spatial_Partition subclasses that run Shoot on an MI355X through
// libhare_hip.so.  Same constructor signatures as the reference classes:
//   Voxel_Grid(Topology[] Model_in, int Domain)                        Voxel_Grid.cs:48
//   Voxel_Grid(Topology[] Model_in, int MaxDomain, int Avg_polys)      Voxel_Grid.cs:128
//   Octree(Topology[] Model_In, int maxDepth, int maxPolygonsPerNode)  "Octree - alt.cs":45
//   KDTree(Topology[] Model_In, int maxDepth, int maxPolygonsPerNode)  KDTree.cs:51
// plus one new member, the batch Shoot, which is where the throughput is.  The single-ray overrides --
// what an unchanged Pachyderm call site reaches -- run hare_shoot_one: the same trace on the calling host
// thread (no GPU round trip per ray, no lock; safe from any number of worker threads, like the reference's
// ThreadID / mailbox-pool design intends), bit-identical to the batch kernels.
//
// The scene is read through public members only: Polygon_Count, this[poly, corner], Polys[p].VertextCT,
// Normal(p), Min, Max -- i.e. AFTER the Topology rounded/merged vertices (Hare_Geometry_Topology.cs:342-377)
// and AFTER Finish_Topology() set Min/Max (:148-167).  Call Finish_Topology() before constructing.
//
// NOT COMPILED IN THE BUILD CONTAINER (no .NET toolchain): see INTEGRATION.md.
using System;
using System.Runtime.InteropServices;

namespace Hare
{
    namespace Geometry
    {
        public abstract class Gpu_Spatial_Partition : Spatial_Partition, IDisposable
        {
            /// <summary>HIP device ordinal new partitions are created on (the constructors keep the reference's
            /// exact signatures, so the device is not a constructor argument).</summary>
            public static int Device = 0;

            /// <summary>Optional: several device ordinals of one node.  A partition constructed while this is set keeps a
            /// replica on each of them and the batch Shoot overloads split their rays over the replicas in contiguous
            /// shards (hare_shoot_batch_sharded): the in-process way to use all GPUs of a node.</summary>
            public static int[] Devices = null;

            /// <summary>Opt-in, default off: reproduce the reference's Ray_ID == 0 rule.  Voxel_Grid.Shoot and KDTree.Shoot skip a
            /// polygon whose mailbox entry equals R.Ray_ID (Voxel_Grid.cs:687-689, KDTree.cs:224-229) and the mailbox starts out all
            /// zero, so a ray with Ray_ID == 0 finds every polygon "already tested" and the reference returns X_Event() -- after
            /// moving an outside origin like for any ray.  The GPU classes keep no mailbox and return the hit; with this switch on,
            /// Shoot(R, ...) and Shoot(Ray[], ...) return X_Event() for rays whose Ray_ID is 0 (Voxel_Grid and KDTree; the live
            /// Octree has no mailbox, "Octree - alt.cs":221-222).  The reference's rule is stateful (a polygon another ray of the
            /// same ThreadID tested since is no longer skipped); this reproduces the fresh-mailbox case.</summary>
            public bool MailboxRayId0 = false;

            protected IntPtr scene = IntPtr.Zero;                 // replica 0 (== scenes[0])
            protected IntPtr[] scenes = new IntPtr[0];            // all replicas, in shard order
            protected abstract int Kind { get; }

            /// <summary>Run a build call on every replica.</summary>
            protected void BuildAll(Func<IntPtr, int> build)
            {
                foreach (IntPtr s in scenes) HareHip.Check(build(s));
            }

            protected Gpu_Spatial_Partition(Topology[] Model_in)
            {
                int[] devices = (Devices != null && Devices.Length > 0) ? (int[])Devices.Clone() : new int[] { Device };
                Model = Model_in;
                var descs = new hare_topology_desc[Model.Length];
                var pins = new GCHandle[Model.Length * 3];
                try
                {
                    for (int m = 0; m < Model.Length; m++)
                    {
                        Topology T = Model[m];
                        int P = T.Polygon_Count;
                        double[] verts = new double[P * 12];
                        int[] nverts = new int[P];
                        double[] normals = new double[P * 3];
                        for (int p = 0; p < P; p++)
                        {
                            int nv = T.Polys[p].VertextCT;
                            nverts[p] = nv;
                            for (int c = 0; c < nv && c < 4; c++)
                            {
                                Point pt = T[p, c];
                                verts[p * 12 + 3 * c + 0] = pt.x;
                                verts[p * 12 + 3 * c + 1] = pt.y;
                                verts[p * 12 + 3 * c + 2] = pt.z;
                            }
                            Vector N = T.Normal(p);
                            normals[p * 3 + 0] = N.dx;
                            normals[p * 3 + 1] = N.dy;
                            normals[p * 3 + 2] = N.dz;
                        }
                        pins[3 * m + 0] = GCHandle.Alloc(verts, GCHandleType.Pinned);
                        pins[3 * m + 1] = GCHandle.Alloc(nverts, GCHandleType.Pinned);
                        pins[3 * m + 2] = GCHandle.Alloc(normals, GCHandleType.Pinned);
                        descs[m].P = P;
                        descs[m].verts = pins[3 * m + 0].AddrOfPinnedObject();
                        descs[m].nverts = pins[3 * m + 1].AddrOfPinnedObject();
                        descs[m].normals = pins[3 * m + 2].AddrOfPinnedObject();
                        descs[m].min0 = T.Min.x; descs[m].min1 = T.Min.y; descs[m].min2 = T.Min.z;
                        descs[m].max0 = T.Max.x; descs[m].max1 = T.Max.y; descs[m].max2 = T.Max.z;
                    }
                    scenes = new IntPtr[devices.Length];
                    for (int k = 0; k < devices.Length; k++)
                        HareHip.Check(HareHip.hare_scene_create(descs, descs.Length, devices[k], out scenes[k]));   // copies everything
                    scene = scenes[0];
                }
                finally
                {
                    foreach (GCHandle h in pins) if (h.IsAllocated) h.Free();
                }
            }

            /// <summary>
            /// Shoot for a whole batch: results[i] is what Shoot(rays[i], top_index, out ..., poly_origin1[i], poly_origin2[i])
            /// returns.  Rays that start outside the grid are moved to the bounding-box entry, like the reference moves R
            /// (AABB_Main.cs:254-257).  Returns the number of hits.
            /// </summary>
            public int Shoot(Ray[] rays, int top_index, X_Event[] results, int[] poly_origin1 = null, int[] poly_origin2 = null)
            {
                int n = rays.Length;
                if (results.Length < n) throw new ArgumentException("results is shorter than rays");
                var r = new hare_ray[n];
                for (int i = 0; i < n; i++)
                {
                    r[i].x = rays[i].x; r[i].y = rays[i].y; r[i].z = rays[i].z;
                    r[i].dx = rays[i].dx; r[i].dy = rays[i].dy; r[i].dz = rays[i].dz;
                }
                var ev = new hare_xevent[n];
                hare_counters ctr;
                HareHip.Check(HareHip.hare_shoot_batch_sharded(scenes, scenes.Length, Kind, top_index, n, r, poly_origin1, poly_origin2,
                                                               HareHip.HARE_SHOOT_WRITEBACK_ORIGIN, ev, out ctr));
                for (int i = 0; i < n; i++)
                {
                    rays[i].x = r[i].x; rays[i].y = r[i].y; rays[i].z = r[i].z;       // F11: the reference mutates R
                    bool id0 = MailboxRayId0 && Kind != HareHip.HARE_KIND_OCTREE && rays[i].Ray_ID == 0;
                    if (id0 && ev[i].hit != 0) ctr.hits--;
                    results[i] = (ev[i].hit != 0 && !id0)
                        ? new X_Event(new Point(ev[i].x, ev[i].y, ev[i].z), ev[i].u, ev[i].v, ev[i].t, ev[i].poly_id)
                        : new X_Event();
                }
                return (int)ctr.hits;
            }

            /// <summary>Zero-copy flavour for callers that keep rays/events in flat arrays (no per-ray objects).</summary>
            public int Shoot(hare_ray[] rays, int top_index, hare_xevent[] results, int[] poly_origin1 = null, int[] poly_origin2 = null, bool moveOrigins = false)
            {
                hare_counters ctr;
                HareHip.Check(HareHip.hare_shoot_batch_sharded(scenes, scenes.Length, Kind, top_index, rays.Length, rays, poly_origin1, poly_origin2,
                                                               moveOrigins ? HareHip.HARE_SHOOT_WRITEBACK_ORIGIN : 0u, results, out ctr));
                return (int)ctr.hits;
            }

            /// <summary>Voxel_Grid only: the batch Shoot with slim result records -- 16 bytes per ray come back over the host link
            /// instead of 56 (+30 % rays per second from host arrays on the measured box).  For a ray that starts inside the grid
            /// (hit == 1) the caller rebuilds what it needs itself: X_Event.t = t, X_Point = origin + direction * t (the reference's
            /// expression, same bits), u = v = 0.  Expand(rays, slim, events) rebuilds every record, moved origins included.</summary>
            public int ShootSlim(hare_ray[] rays, int top_index, hare_slim_event[] results, int[] poly_origin1 = null, int[] poly_origin2 = null)
            {
                if (Kind != HareHip.HARE_KIND_VOXEL) throw new InvalidOperationException("slim records of the trees carry u, v (32 bytes): use Shoot");
                if (results.Length < rays.Length) throw new ArgumentException("results is shorter than rays");
                hare_counters ctr;
                HareHip.Check(HareHip.hare_shoot_batch_sharded_slim(scenes, scenes.Length, Kind, top_index, rays.LongLength, rays, poly_origin1, poly_origin2,
                                                                    HareHip.HARE_SHOOT_SLIM_EVENTS, results, out ctr));
                return (int)ctr.hits;
            }

            /// <summary>Slim records (of ShootSlim on these rays) to full records, byte-identical to what Shoot returns.</summary>
            public void Expand(hare_ray[] rays, hare_slim_event[] slim, hare_xevent[] events)
            {
                HareHip.Check(HareHip.hare_expand_events(scene, Kind, rays.LongLength, rays, slim, events));
            }

#if NET7_0_OR_GREATER
            /// <summary>Span flavour (net7.0 target): rays/events may live in any contiguous memory -- a slice of a larger
            /// array, native memory, a stackalloc -- and are pinned only for the duration of the call.</summary>
            public unsafe int Shoot(Span<hare_ray> rays, int top_index, Span<hare_xevent> results,
                                    ReadOnlySpan<int> poly_origin1 = default, ReadOnlySpan<int> poly_origin2 = default, bool moveOrigins = false)
            {
                if (results.Length < rays.Length) throw new ArgumentException("results is shorter than rays");
                if (!poly_origin1.IsEmpty && poly_origin1.Length < rays.Length) throw new ArgumentException("poly_origin1 is shorter than rays");
                if (!poly_origin2.IsEmpty && poly_origin2.Length < rays.Length) throw new ArgumentException("poly_origin2 is shorter than rays");
                hare_counters ctr;
                fixed (hare_ray* r = rays)
                fixed (hare_xevent* e = results)
                fixed (int* e1 = poly_origin1)      // an empty span pins to null == "no exclusions"
                fixed (int* e2 = poly_origin2)
                    HareHip.Check(HareHip.hare_shoot_batch_sharded(scenes, scenes.Length, Kind, top_index, rays.Length, r, e1, e2,
                                                                   moveOrigins ? HareHip.HARE_SHOOT_WRITEBACK_ORIGIN : 0u, e, &ctr));
                return (int)ctr.hits;
            }
#endif

            public override bool Shoot(Ray R, int top_index, out X_Event Ret_event)
            {
                return Shoot(R, top_index, out Ret_event, -1, -1);
            }

            public override bool Shoot(Ray R, int top_index, out X_Event Ret_event, int poly_origin1, int poly_origin2 = -1)
            {
                hare_ray r;
                r.x = R.x; r.y = R.y; r.z = R.z; r.dx = R.dx; r.dy = R.dy; r.dz = R.dz;
                hare_xevent ev;
                HareHip.Check(HareHip.hare_shoot_one(scene, Kind, top_index, ref r, poly_origin1, poly_origin2, out ev));
                R.x = r.x; R.y = r.y; R.z = r.z;                                  // F11: the reference mutates R
                if (MailboxRayId0 && Kind != HareHip.HARE_KIND_OCTREE && R.Ray_ID == 0) { Ret_event = new X_Event(); return false; }
                Ret_event = ev.hit != 0 ? new X_Event(new Point(ev.x, ev.y, ev.z), ev.u, ev.v, ev.t, ev.poly_id) : new X_Event();
                return Ret_event.Hit;
            }

            /// <summary>Occlusion predicate for a batch (harness-defined; the reference has no any-hit call, its seam is
            /// Spatial_Partition.cs:32-33): occluded[i] = the closest hit of rays[i] exists and lies before t_max[i]
            /// (t_max null: any hit).  Runs on the GPU(s) like the batch Shoot, sharded over Devices.</summary>
            public bool[] Occluded(hare_ray[] rays, int top_index, double[] t_max = null, int[] poly_origin1 = null, int[] poly_origin2 = null)
            {
                var occ = new int[rays.Length];
                hare_counters ctr;
                // every replica takes a contiguous shard, like the batch Shoot; no events: the kernels stop each ray once its flag is decided
                HareHip.Check(HareHip.hare_occluded_batch_sharded(scenes, scenes.Length, Kind, top_index, rays.LongLength, rays, poly_origin1, poly_origin2,
                                                                  t_max, 0u, occ, null, out ctr));
                var res = new bool[rays.Length];
                for (int i = 0; i < res.Length; i++) res[i] = occ[i] != 0;
                return res;
            }

            /// <summary>The bounce loop a Pachyderm-style caller runs per ray with the reference -- Shoot, reflect about
            /// Model[top_index].Normal(Poly_id) (Hare_Geometry_Polygons.cs:161-171), Shoot again with poly_origin1 = the
            /// polygon just hit (Spatial_Partition.cs:33) -- for a whole batch and `bounces` casts in ONE native call: the rays
            /// stay on the GPU(s) for all casts, only events come back.  events_all receives bounces x rays.Length records,
            /// cast-major (cast b of ray i at [b * rays.Length + i]); a ray that missed in an earlier cast holds the miss record
            /// (hit == 0, poly_id == -1) from then on.  Returns the number of hits over all casts; per_cast (optional,
            /// `bounces` entries) receives the rays alive and the hits of every cast.</summary>
            public long Bounce(hare_ray[] rays, int top_index, int bounces, hare_xevent[] events_all, int[] poly_origin1 = null,
                               int[] poly_origin2 = null, hare_counters[] per_cast = null)
            {
                if (bounces < 1) throw new ArgumentException("bounces must be at least 1");
                if (events_all == null || events_all.LongLength < (long)bounces * rays.LongLength) throw new ArgumentException("events_all must hold bounces x rays.Length records");
                if (per_cast != null && per_cast.Length < bounces) throw new ArgumentException("per_cast is shorter than bounces");
                hare_counters ctr;
                HareHip.Check(HareHip.hare_bounce_batch_sharded(scenes, scenes.Length, Kind, top_index, rays.LongLength, rays, poly_origin1, poly_origin2,
                                                                bounces, 0u, events_all, null, out ctr, per_cast));
                return (long)ctr.hits;
            }

            /// <summary>The same on managed objects: result[b][i] is the X_Event of ray i in cast b (X_Event() once the ray has
            /// left the model).  rays[] is not modified.</summary>
            public X_Event[][] Bounce(Ray[] rays, int top_index, int bounces)
            {
                int n = rays.Length;
                var r = new hare_ray[n];
                for (int i = 0; i < n; i++)
                {
                    r[i].x = rays[i].x; r[i].y = rays[i].y; r[i].z = rays[i].z;
                    r[i].dx = rays[i].dx; r[i].dy = rays[i].dy; r[i].dz = rays[i].dz;
                }
                var ev = new hare_xevent[(long)bounces * n];
                Bounce(r, top_index, bounces, ev);
                var res = new X_Event[bounces][];
                for (int b = 0; b < bounces; b++)
                {
                    res[b] = new X_Event[n];
                    for (int i = 0; i < n; i++)
                    {
                        hare_xevent e = ev[(long)b * n + i];
                        res[b][i] = e.hit != 0 ? new X_Event(new Point(e.x, e.y, e.z), e.u, e.v, e.t, e.poly_id) : new X_Event();
                    }
                }
                return res;
            }

            /// <summary>Diagnostics / A-B switch on every replica (hare_scene_set_option), e.g. SetOption("voxel_kernel", 2).</summary>
            public void SetOption(string name, long value)
            {
                foreach (IntPtr s in scenes) HareHip.Check(HareHip.hare_scene_set_option(s, name, value));
            }

            /// <summary>An option of replica 0 read back, or "voxel_tight_bytes" / "octree_scratch_bytes" (hare_scene_get_option).</summary>
            public long GetOption(string name)
            {
                long v;
                HareHip.Check(HareHip.hare_scene_get_option(scene, name, out v));
                return v;
            }

            void Release()
            {
                for (int k = 0; k < scenes.Length; k++)
                    if (scenes[k] != IntPtr.Zero) { HareHip.hare_scene_destroy(scenes[k]); scenes[k] = IntPtr.Zero; }
                scene = IntPtr.Zero;
            }

            public void Dispose()
            {
                Release();
                GC.SuppressFinalize(this);
            }

            ~Gpu_Spatial_Partition() { Release(); }
        }

        /// <summary>Voxel_Grid on the GPU (Voxel_Grid.cs).</summary>
        public class Gpu_Voxel_Grid : Gpu_Spatial_Partition
        {
            protected override int Kind { get { return HareHip.HARE_KIND_VOXEL; } }
            hare_voxel_info info;

            /// <summary>Voxel_Grid(Topology[] Model_in, int Domain) -- Voxel_Grid.cs:48</summary>
            public Gpu_Voxel_Grid(Topology[] Model_in, int Domain) : base(Model_in)
            {
                BuildAll(sc => HareHip.hare_voxel_build(sc, Domain));
                Refresh();
            }

            /// <summary>Voxel_Grid(Topology[] Model_in, int MaxDomain, int Avg_polys) -- Voxel_Grid.cs:128</summary>
            public Gpu_Voxel_Grid(Topology[] Model_in, int MaxDomain, int Avg_polys) : base(Model_in)
            {
                BuildAll(sc => HareHip.hare_voxel_build_adaptive(sc, MaxDomain, Avg_polys));
                Refresh();
            }

            void Refresh()
            {
                HareHip.Check(HareHip.hare_voxel_get_info(scene, out info));
                Char_Step = info.char_step;                                   // Voxel_Grid.cs:90
            }

            public double Xdim { get { return info.box_dims0; } }            // Voxel_Grid.cs:763-783
            public double Ydim { get { return info.box_dims1; } }
            public double Zdim { get { return info.box_dims2; } }
            public Point MinPt { get { return new Point(info.obox_min0, info.obox_min1, info.obox_min2); } }   // :785-791

            public void PointInVoxel(Point Pt, out int X, out int Y, out int Z)   // Voxel_Grid.cs:322-327
            {
                X = (int)Math.Floor((Pt.x - info.obox_min0) / info.voxel_dims0);
                Y = (int)Math.Floor((Pt.y - info.obox_min1) / info.voxel_dims1);
                Z = (int)Math.Floor((Pt.z - info.obox_min2) / info.voxel_dims2);
            }

            public int VoxelCode(int X, int Y, int Z) { return info.ct * info.ct * Z + info.ct * X + Y; }     // :264-267

            public int PointInVoxel(Point Pt)                                                                 // :329-332
            {
                int X, Y, Z;
                PointInVoxel(Pt, out X, out Y, out Z);
                return VoxelCode(X, Y, Z);
            }

            public void VoxelDecode(int Code, out int X, out int Y, out int Z)                                // :256-262
            {
                int XYTot = info.ct * info.ct;
                Z = (int)Math.Floor((double)(Code / XYTot));
                Code -= Z * XYTot;
                Y = (int)Math.Floor((double)(Code / info.ct));
                X = Code - Y * info.ct;
            }

            System.Collections.Generic.List<int>[,,,] voxel_inv;
            /// <summary>The reference's public field `List&lt;int&gt;[,,,] Voxel_Inv` (Voxel_Grid.cs:33): polygon indices per
            /// voxel and topology, ascending.  Materialised on first use from the library's CSR lists (hare_voxel_get_lists);
            /// the kernels never read it.</summary>
            public System.Collections.Generic.List<int>[,,,] Voxel_Inv
            {
                get
                {
                    if (voxel_inv != null) return voxel_inv;
                    int ct = info.ct, M = Model.Length;
                    var inv = new System.Collections.Generic.List<int>[ct, ct, ct, M];
                    var start = new uint[ct * ct * ct + 1];
                    for (int m = 0; m < M; m++)
                    {
                        HareHip.Check(HareHip.hare_voxel_get_lists(scene, m, start, null));
                        var items = new int[Math.Max(1, (int)start[start.Length - 1])];
                        HareHip.Check(HareHip.hare_voxel_get_lists(scene, m, start, items));
                        for (int x = 0; x < ct; x++) for (int y = 0; y < ct; y++) for (int z = 0; z < ct; z++)
                        {
                            int c = (x * ct + y) * ct + z;
                            var l = new System.Collections.Generic.List<int>((int)(start[c + 1] - start[c]));
                            for (uint k = start[c]; k < start[c + 1]; k++) l.Add(items[k]);
                            inv[x, y, z, m] = l;
                        }
                    }
                    voxel_inv = inv;
                    return inv;
                }
            }
        }

        /// <summary>Octree on the GPU ("Octree - alt.cs").</summary>
        public class Gpu_Octree : Gpu_Spatial_Partition
        {
            protected override int Kind { get { return HareHip.HARE_KIND_OCTREE; } }
            public Gpu_Octree(Topology[] Model_In, int maxDepth, int maxPolygonsPerNode) : base(Model_In)
            {
                BuildAll(sc => HareHip.hare_octree_build(sc, maxDepth, maxPolygonsPerNode));
            }
        }

        /// <summary>KDTree on the GPU (KDTree.cs).</summary>
        public class Gpu_KDTree : Gpu_Spatial_Partition
        {
            protected override int Kind { get { return HareHip.HARE_KIND_KDTREE; } }
            public Gpu_KDTree(Topology[] Model_In, int maxDepth, int maxPolygonsPerNode) : base(Model_In)
            {
                BuildAll(sc => HareHip.hare_kdtree_build(sc, maxDepth, maxPolygonsPerNode));
            }
        }
    }
}
